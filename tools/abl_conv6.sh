# conv_gemm6's K loop with parts left out (timing only, wrong results): which resource bounds it?  GPU box.
# usage: bash tools/abl_conv6.sh [ops] [abl values]      (PPMS_CONV6_ABL: tools/gen_conv6_asm.py; phase times + loop clock: tools/conv6_phase_probe.py)
OPS=${1:-zr1_0_x,fh1,q1_x,zr3_x}
trap 'env -u PPMS_CONV6_ABL -u PPMS_BUILD_DEFINES python tools/gen_conv6_asm.py > /dev/null; env -u PPMS_BUILD_DEFINES python -m ppmstereo_amd.build > /dev/null 2>&1' EXIT
export PPMS_BUILD_DEFINES="-DPPMS_CONV6_TIMING"
for a in ${2:-0 1 2 3 4}; do
  PPMS_CONV6_ABL=$a python tools/gen_conv6_asm.py > /dev/null || exit 1
  python -m ppmstereo_amd.build > /dev/null 2>&1 || exit 1
  echo "--- PPMS_CONV6_ABL=$a"
  python tools/conv6_phase_probe.py $OPS 2>&1 | grep -v "amdgpu.ids\|ab_switches"
done
