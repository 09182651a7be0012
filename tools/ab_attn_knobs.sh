# stand-alone 1/4-scale attention call (tools/attn_probe.py, random operands) for generator settings, e.g.
#   bash tools/ab_attn_knobs.sh "PPMS_ATTN_DSLOT=18,22,50,58" "PPMS_ATTN_DSLOT=2,6,42,46"
# whatever happens below (a failed build, a timeout, ^C), the committed default header comes back: the library's digest includes it
trap 'env -u PPMS_CONV5_ABL -u PPMS_CONV5_ACC -u PPMS_ATTN_ABL -u PPMS_ATTN_DSLOT python tools/gen_attn_asm.py > /dev/null' EXIT
for r in 1 2; do
for a in "$@"; do
  env $a python tools/gen_attn_asm.py > /dev/null && python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1 &&
  echo "$a: $(timeout -k 10 100 python tools/attn_probe.py 15 1 2>&1 | tail -1)" || exit 1
done
done
