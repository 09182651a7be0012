# timing experiments on the hand-scheduled attention loop: regenerates attn64_asm.h with parts dropped (results are wrong), rebuilds,
# times tools/attn_probe.py.  usage (on the GPU box): bash tools/abl_attn.sh 0 1 2 3 ...
for a in "$@"; do
  PPMS_ATTN_ABL=$a python tools/gen_attn_asm.py > /dev/null && python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1 &&
  echo "ABL=$a: $(timeout -k 10 100 python tools/attn_probe.py 7 1 2>&1 | tail -1)"
done
PPMS_ATTN_ABL=0 python tools/gen_attn_asm.py > /dev/null
