for v in ${RINGS:-"PPMS_ATTN_RING=4" "PPMS_ATTN_RING=6" "PPMS_ATTN_RING=4" "PPMS_ATTN_RING=6"}; do
  env $v python tools/gen_attn_asm.py > /dev/null && python -c "from ppmstereo_amd import build; build.build(verbose=False)" > /dev/null 2>&1 &&
  python bench.py --steps 20 --no-encoders --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=[d[k] for k in ('roofline','roofline_2') if 'per_scale' in d[k]][0]; print('$v', 'ms_per_step', d['ms_per_step'], 'attn 1/4', r['per_scale']['1/4']['avg_ms'], r['per_scale']['1/4']['tflops'])"
done
python tools/gen_attn_asm.py > /dev/null
