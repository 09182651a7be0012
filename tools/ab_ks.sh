# A/B on one box: conv_gemm6's K-split form of M = 128 (default build) against a -DCONV6_KSPLIT=0 build, alternating.  Build the second library first (here,
# no GPU needed):  PPMS_BUILD_DEFINES="-DCONV6_KSPLIT=0" python -m ppmstereo_amd.build && mkdir -p ab_libs && cp ppmstereo_amd/libppms.so ab_libs/libppms_noks.so
#                  && python -m ppmstereo_amd.build        (ab_libs/ is git-ignored and travels to the GPU box)
mkdir -p gpurun_out/abks
run() { tag=$1; shift; env "$@" timeout -k 10 200 python tools/ab_bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-encoders > gpurun_out/abks/$tag.json 2> gpurun_out/abks/$tag.err; python -c "
import json; p=json.loads(open('gpurun_out/abks/$tag.json').read().strip().splitlines()[-1]); po=p['roofline']['per_op']; print('$tag', p['ms_per_step'], p['ms_per_step_median'], p['roofline']['achieved'], p['roofline']['total_ms_per_step'], {k.split(':')[1]:round(v['avg_ms']*1e3,1) for k,v in po.items() if k.split(':')[1] in ('q1_x','q2_x','unc0','zr1_2')})"; }
run ks A=1
run noks PPMS_LIB=ab_libs/libppms_noks.so
run ks2 A=1
run noks2 PPMS_LIB=ab_libs/libppms_noks.so
