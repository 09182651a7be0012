# Every measured artefact of round 6 with ONE build (GPU box): gpurun_out/r06/ -> copied into profiles/ as r06_*.
# usage: bash tools/r06_profiles.sh [part ...]     parts: bench trace traffic mfma pmc parity tests big wholecall versions   (default: all)
set -o pipefail
parts=${@:-bench trace traffic mfma pmc parity tests big wholecall versions}
out=gpurun_out/r06
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for part in $parts; do
  echo "== $part"
  case $part in
    bench) timeout -k 10 600 python bench.py > $out/final_bench.json 2> $out/final_bench.err || exit 1
           python -c "import json; d=json.loads(open('$out/final_bench.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'whole call', d['whole_call_ms'], [(r['kernel'][:14], r['achieved'], r['frac'], r['total_ms_per_step']) for r in (d['roofline'], d['roofline_2'])], d['roofline_3']['ms_per_step_by_scale'])" ;;
    trace) bash tools/gpu_prof.sh r06/trace > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
           tail -8 $out/trace.log | cut -c1-160; du -sh $out/trace ;;
    traffic) bash tools/traffic_pmc.sh r06/traffic > $out/traffic.log 2>&1 || { tail -5 $out/traffic.log; exit 1; }
             tail -3 $out/traffic.log; find $out/traffic -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + ;;       # (keep the JSON summaries, drop the raw counter dumps: gpurun_out/ returns <= 64 MiB)
    mfma) bash tools/mfma_util.sh r06/mfma > $out/mfma.log 2>&1 || { tail -5 $out/mfma.log; exit 1; }
          tail -3 $out/mfma.log; find $out/mfma -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + ;;
    pmc) bash tools/conv6_pmc.sh r06/conv6pmc zr1_0_x conv6_kernel > $out/conv6_pmc.log 2>&1 || { tail -5 $out/conv6_pmc.log; exit 1; }
         tail -4 $out/conv6_pmc.log; find $out/conv6pmc -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + ;;
    parity) timeout -k 10 900 python -m pytest tests/test_gpu_zz_full_configs.py -x -q -s > $out/parity.log 2>&1; echo "rc=$?" >> $out/parity.log; tail -3 $out/parity.log ;;
    tests) timeout -k 10 900 python -m pytest tests -q -m gpu > $out/gpu_pytest.log 2>&1; echo "rc=$?" >> $out/gpu_pytest.log; tail -4 $out/gpu_pytest.log ;;
    big) timeout -k 10 300 python bench.py --T 5 --H 736 --W 1280 --iters 20 --steps 3 --warmup 1 --no-cpu-baseline --no-encoders > $out/bench_cfg3_736x1280_iters20.json 2> $out/cfg3.err || exit 1
         timeout -k 10 300 python bench.py --T 40 --H 320 --W 512 --iters 20 --steps 3 --warmup 1 --no-cpu-baseline --no-encoders > $out/bench_T40_320x512_iters20_single_gpu.json 2> $out/t40.err || exit 1
         python -c "import json; [print(f, json.loads(open('$out/'+f).read().strip().splitlines()[-1])['ms_per_step']) for f in ('bench_cfg3_736x1280_iters20.json','bench_T40_320x512_iters20_single_gpu.json')]" ;;
    wholecall) R=$PWD; (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/$out/wc -o trace -- /usr/bin/python3 $R/tools/whole_call_probe.py > $R/$out/whole_call_probe.txt 2>&1) || { tail -5 $out/whole_call_probe.txt; exit 1; }
               python tools/whole_call_timeline.py $(find $out/wc -name "*kernel_trace.csv" | head -1) > $out/whole_call_timeline.txt 2>&1; rm -rf $out/wc; head -8 $out/whole_call_timeline.txt | cut -c1-180 ;;
    versions) for g in "5 320 512" "5 736 1280" "40 320 512"; do python tools/kernel_versions.py $g 2>/dev/null; done > $out/kernel_versions.txt; grep -c conv_gemm $out/kernel_versions.txt ;;
  esac
done
