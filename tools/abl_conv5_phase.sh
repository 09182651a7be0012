# conv5 phase times for ablation builds (timing only, wrong results).  Each argument: "[PPMS_CONV5_ABL=n] [DEF=-D...]"
#   PPMS_CONV5_ABL (tools/gen_conv5_asm.py): 1 drops the LDS fragment reads, 2 the weight-fragment loads, 4 the MFMAs;
#   -DCONV5_NOSYNC=1 drops wait + barrier + DMA at the window switches, 2 the DMA, 3 the barrier;  -DCONV5_ABL_A=1 reloads the first step's weights.
# usage (GPU box): bash tools/abl_conv5_phase.sh "" "DEF=-DCONV5_NOSYNC=1" "PPMS_CONV5_ABL=3 DEF=-DCONV5_NOSYNC=1" ...
# whatever happens below (a failed build, a timeout, ^C), the committed default header comes back: the library's digest includes it
trap 'env -u PPMS_CONV5_ABL -u PPMS_CONV5_ACC -u PPMS_ATTN_ABL -u PPMS_ATTN_DSLOT python tools/gen_conv5_asm.py > /dev/null' EXIT
for a in "$@"; do
  DEF=""; ABL=0
  for kv in $a; do case $kv in DEF=*) DEF="${kv#DEF=}";; PPMS_CONV5_ABL=*) ABL="${kv#PPMS_CONV5_ABL=}";; esac; done
  export PPMS_BUILD_DEFINES="-DPPMS_CONV5_TIMING $DEF"
  PPMS_CONV5_ABL=$ABL python tools/gen_conv5_asm.py > /dev/null && python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1 || exit 1
  echo "== '$a'"
  timeout -k 10 200 python tools/conv5_phase_probe.py ${OPS:-zr1_0,zr2,fh1,q1,unc0} 2>&1 | grep -v "amdgpu.ids" | grep "workgroups\|phase means"
done
