# PMC counters of ONE conv op of the 1/4-scale engine as the loop runs it (counter collection only; interpreter directly behind `--`).
# usage (GPU box): bash tools/conv6_pmc.sh <tag> [op] [kernel-name substring]   -> gpurun_out/<tag>/pmc_pass{1,2}.txt
set -o pipefail
tag=${1:-conv6pmc}; op=${2:-zr1_0_x}; kern=${3:-conv6_kernel}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
ROOT=$PWD
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
    --output-format csv -d $out/p1 -o pmc -- /usr/bin/python3 $ROOT/tools/conv_pmc_probe.py $op 4 > $out/p1.log 2>&1 || { tail -5 $out/p1.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM \
    --output-format csv -d $out/p2 -o pmc -- /usr/bin/python3 $ROOT/tools/conv_pmc_probe.py $op 4 > $out/p2.log 2>&1 || { tail -5 $out/p2.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAIT_INST_ANY \
    --output-format csv -d $out/p3 -o pmc -- /usr/bin/python3 $ROOT/tools/conv_pmc_probe.py $op 4 > $out/p3.log 2>&1 || { tail -5 $out/p3.log; exit 1; }
cd $ROOT
for p in p1 p2 p3; do python tools/pmc_summary.py $(find $out/$p -name "*counter_collection.csv" | head -1) $kern > $out/pmc_$p.txt; cat $out/pmc_$p.txt; done
