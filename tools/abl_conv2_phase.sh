# conv2 phase times at a small scale for ablation builds (timing only).  usage (GPU box): bash tools/abl_conv2_phase.sh <scale> "" "-DCONV2_ABL_A=1" ...
sc=$1; shift
for d in "$@"; do
  export PPMS_BUILD_DEFINES="-DPPMS_CONV2_TIMING $d"
  python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1 || exit 1
  echo "== scale 1/$sc, defines '$d'"
  timeout -k 10 200 python tools/conv2_phase_probe.py $sc 2>&1 | grep -v "amdgpu.ids"
done
