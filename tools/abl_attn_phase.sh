# per-workgroup loop time of the 64-query attention kernel for generator variants (PPMS_ATTN_ABL drops parts of the loop -- wrong results,
# timing only; the probe's operands are zeros: cycles, not clocks).  DEF = extra build defines.
# usage (on the GPU box): bash tools/abl_attn_phase.sh "PPMS_ATTN_ABL=0" "PPMS_ATTN_ABL=1 DEF=-DPPMS_ATTN_NOSYNC" ...
# whatever happens below (a failed build, a timeout, ^C), the committed default header comes back: the library's digest includes it
trap 'env -u PPMS_CONV5_ABL -u PPMS_CONV5_ACC -u PPMS_ATTN_ABL -u PPMS_ATTN_DSLOT python tools/gen_attn_asm.py > /dev/null' EXIT
for a in "$@"; do
  DEF=""; for kv in $a; do case $kv in DEF=*) DEF="${kv#DEF=}";; esac; done
  export PPMS_BUILD_DEFINES="-DPPMS_ATTN_TIMING $DEF"
  env $a python tools/gen_attn_asm.py > /dev/null && python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1 &&
  echo "$a: $(timeout -k 10 100 python tools/attn_phase_probe.py 2>&1 | grep 'per workgroup')" || exit 1
done
