# kernel trace of bench.py only (no tests).  usage: bash tools/gpu_prof.sh <tag> [bench args]
set -o pipefail
tag=${1:-prof}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 &&
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encoders "$@" > $out/bench.json 2> $out/bench.err &&
python -c "import json,sys; d=json.loads(open('$out/bench.json').readline()); print('ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], [ (r['kernel'][:12], r['achieved'], r['total_ms_per_step']) for r in (d['roofline'], d['roofline_2']) if r])" &&
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/prof -o trace -- /usr/bin/python3 $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-encoders "$@" > $OLDPWD/$out/bench_prof.json 2> $OLDPWD/$out/prof.err) &&
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv &&
find $out/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace.csv &&
python tools/trace_torch_ops.py $out/kernel_trace.csv 2 > $out/torch_ops.txt; python tools/trace_iter.py $out/kernel_trace.csv 4 > $out/iter4.txt && python tools/trace_iter.py $out/kernel_trace.csv 8 > $out/iter8.txt && python tools/trace_iter.py $out/kernel_trace.csv 16 > $out/iter16.txt && rm -rf $out/prof $out/kernel_trace.csv
head -14 $out/kernel_stats.csv | cut -c1-150
