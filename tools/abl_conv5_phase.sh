# conv5 phase times for ablation builds (timing only).  usage (GPU box): bash tools/abl_conv5_phase.sh "" "-DCONV5_NOSYNC=1" ...
for d in "$@"; do
  export PPMS_BUILD_DEFINES="-DPPMS_CONV5_TIMING $d"
  python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1 || exit 1
  echo "== defines '$d'"
  timeout -k 10 200 python tools/conv5_phase_probe.py zr1_0,zr2,fh1,q1,unc0 2>&1 | grep -v "amdgpu.ids" | grep "workgroups\|phase means"
done
