# HBM-side traffic of the two dominant kernels, as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (counter collection only, the interpreter binary directly behind `--`), then tools/traffic_summary.py
# applies the gfx950 correction (FETCH_SIZE x2 for 16-B/lane streams) and writes profiles-ready JSON.
# usage (GPU box): bash tools/traffic_pmc.sh <tag>
set -o pipefail
tag=${1:-traffic}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 &&
cd /tmp &&
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/attn_$c -o pmc -- /usr/bin/python3 $OLDPWD/tools/attn_probe.py 4 1 > $out/attn_$c.log 2>&1 &&
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/conv_$c -o pmc -- /usr/bin/python3 $OLDPWD/tools/conv_pmc_probe.py zr1_0_x 4 > $out/conv_$c.log 2>&1 &&
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/corr_$c -o pmc -- /usr/bin/python3 $OLDPWD/tools/corr_probe.py 5 80 128 > $out/corr_$c.log 2>&1 || exit 1
done
cd $OLDPWD &&
python tools/traffic_summary.py $out
