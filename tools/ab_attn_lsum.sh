# A/B of the attention's softmax-denominator form on the GPU box: accuracy (the iters=10 fixtures, config 2 against the oracle, the
# structured-video decomposition) and speed (bench.py) with PPMS_ATTN_LSUM=add / mfma / PPMS_ATTN_DOT=1.   usage: bash tools/ab_attn_lsum.sh
for v in "PPMS_ATTN_LSUM=add" "PPMS_ATTN_LSUM=add PPMS_ATTN_DOT=1" "PPMS_ATTN_LSUM=mfma"; do
  echo "=== $v"
  env $v python tools/gen_attn_asm.py > /dev/null && python -c "from ppmstereo_amd import build; build.build(verbose=False)" &&
  python -m pytest tests/test_gpu_block.py tests/test_gpu_zz_full_configs.py -q -s -k "iters10 or ten_iterations or full_iteration or decomposition" 2>&1 | grep -E "final disparity|iteration 9|structured video, oracle|structured video, all|config 2, final|passed|failed" &&
  python bench.py --steps 10 --no-encoders --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=[d[k] for k in ('roofline','roofline_2') if 'per_scale' in d[k]][0]; print('ms_per_step', d['ms_per_step'], 'attn 1/4', r['per_scale']['1/4'])"
done
python tools/gen_attn_asm.py > /dev/null
