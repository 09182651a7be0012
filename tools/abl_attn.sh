# timing experiments on the hand-scheduled attention loop: regenerates attn64_asm.h with the given environment (PPMS_ATTN_ABL drops
# parts -- wrong results; PPMS_ATTN_PK=0/1 scalar / packed fp32 softmax ops), rebuilds, times tools/attn_probe.py.
# usage (on the GPU box): bash tools/abl_attn.sh "PPMS_ATTN_PK=0" "PPMS_ATTN_PK=1 PPMS_ATTN_ABL=16" ...
for a in "$@"; do
  env $a python tools/gen_attn_asm.py > /dev/null && python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1 &&
  echo "$a: $(timeout -k 10 100 python tools/attn_probe.py 9 1 2>&1 | tail -1)"
done
python tools/gen_attn_asm.py > /dev/null
