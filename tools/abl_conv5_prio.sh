# conv5 with a static issue priority for one of the two waves of a SIMD (experiment builds): bash tools/abl_conv5_prio.sh 0 1 2 3
trap 'unset PPMS_BUILD_DEFINES; python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1' EXIT
for a in "$@"; do
  export PPMS_BUILD_DEFINES="-DCONV5_PRIO=$a"
  python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1 || exit 1
  echo "== CONV5_PRIO=$a"
  timeout -k 10 200 python tools/conv_probe.py ${OPS:-zr1_0,zr2,fh1,q1,unc0,final_0} 30 2>&1 | grep -v "amdgpu.ids"
done
