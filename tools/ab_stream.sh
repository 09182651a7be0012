# A/B of the register-streamed small-map kernel inside the clip (GPU box): bash tools/ab_stream.sh <outdir>
out=${1:-gpurun_out/ab_stream}; mkdir -p $out
for v in 0 1 all 0 1; do
  PPMS_STREAM=$v timeout -k 10 200 python tools/ab_bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encoders > $out/bench_$v.json 2> $out/bench_$v.err || exit 1
  python - $out/bench_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print("stream=%s  ms/step %.3f  median %.3f  min %.3f   by scale %s  launches %s" % (sys.argv[2], d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"],
      d["roofline_3"]["ms_per_step_by_scale"], d["roofline_3"]["launches_per_step"]))
PY
done
