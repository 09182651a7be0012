# MFMA-pipe utilisation of the two dominant kernels from PMC counters (counter collection only; interpreter directly behind `--`):
#   SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs), GRBM_GUI_ACTIVE (summed over the 8 XCDs), SQ_BUSY_CYCLES.
# usage (GPU box): bash tools/mfma_util.sh <tag>   -> gpurun_out/<tag>/{conv,attn}_pmc.txt
set -o pipefail
tag=${1:-mfma}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 &&
cd /tmp &&
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $out/conv -o pmc -- /usr/bin/python3 $OLDPWD/tools/conv_pmc_probe.py zr1_0_x 4 > $out/conv.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $out/attn -o pmc -- /usr/bin/python3 $OLDPWD/tools/attn_probe.py 4 1 > $out/attn.log 2>&1
cd $OLDPWD
python tools/pmc_summary.py $(find $out/conv -name "*counter_collection.csv" | head -1) conv6_kernel > $out/conv_pmc.txt
python tools/pmc_summary.py $(find $out/attn -name "*counter_collection.csv" | head -1) mem_attn64 > $out/attn_pmc.txt
cat $out/conv_pmc.txt $out/attn_pmc.txt
