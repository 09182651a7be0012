# One GPU-box session: parity tests, bench line, kernel trace.  usage: bash tools/gpu_check.sh <tag> [pytest-args]
# (outputs under gpurun_out/<tag>/; steps are chained with && so that a failed / killed GPU step starts no further one)
set -o pipefail
tag=${1:-run}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 &&
timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -eq 0 ] &&
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err &&
cat $out/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'roof', d['roofline']['achieved'] if d['roofline'] else None, d['roofline_2']['achieved'] if d['roofline_2'] else None)" &&
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/prof -o trace -- /usr/bin/python3 $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OLDPWD/$out/bench_prof.json 2> $OLDPWD/$out/prof.err) &&
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv &&
find $out/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace.csv &&
head -12 $out/kernel_stats.csv | cut -c1-160 &&
python tools/trace_iter.py $out/kernel_trace.csv 4 > $out/iter4.txt && python tools/trace_iter.py $out/kernel_trace.csv 8 > $out/iter8.txt && python tools/trace_iter.py $out/kernel_trace.csv 16 > $out/iter16.txt && rm -rf $out/prof
