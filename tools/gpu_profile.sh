#!/bin/bash
# One GPU-box pass that produces everything profiles/ keeps for a code state:
#   tools/gpu_profile.sh <tag> [batch]
# -> gpurun_out/<tag>_bench.json/.log, <tag>_stats (rocprofv3 --kernel-trace --stats), <tag>_pmc{A,B,C} (SQ counters),
#    <tag>_fetch / <tag>_write (HBM bytes).  Counters are collected in their own runs, program directly after `--`.
tag=$1; batch=${2:-131072}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 4 --warmup 2 --batch $batch --no-cpu --no-profile --ref-batch 0 --min-seconds 0"
python3 bench.py --batch $batch > $out/${tag}_bench.json 2> $out/${tag}_bench.log
cp bench_extras.json $out/${tag}_bench_extras.json   # the full record of THIS run (the later runs overwrite bench_extras.json)
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o ${tag}_stats -- python3 bench.py --steps 20 --warmup 5 --batch $batch --no-cpu --no-profile --ref-batch 0 --min-seconds 0 > $out/${tag}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/${tag}_pmcA -o ${tag}_pmcA --output-format csv -- $BENCH > $out/${tag}_pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES -d $out/${tag}_pmcB -o ${tag}_pmcB --output-format csv -- $BENCH > $out/${tag}_pmcB.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES -d $out/${tag}_pmcC -o ${tag}_pmcC --output-format csv -- $BENCH > $out/${tag}_pmcC.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_fetch -o ${tag}_fetch --output-format csv -- $BENCH > $out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_write -o ${tag}_write --output-format csv -- $BENCH > $out/${tag}_write.log 2>&1
find $out/${tag}_* -name '*.csv' | head -40
# ---- round 3 additions: stand-alone SpMM traffic (both kernels), sr25 sweep, other configs, MNIST-75
SPMM="python3 tools/bench_spmm.py"
$SPMM > $out/${tag}_spmm.json 2> $out/${tag}_spmm.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_spmm_fetch -o ${tag}_spmm_fetch --output-format csv -- $SPMM > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_spmm_write -o ${tag}_spmm_write --output-format csv -- $SPMM > /dev/null 2>&1
python3 tools/bench_sr25_sweep.py > $out/${tag}_sr25_sweep.jsonl 2> $out/${tag}_sr25_sweep.log
SWEEP="python3 tools/bench_sr25_sweep.py --S 48 --nodes 500000"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_sr25_fetch -o ${tag}_sr25_fetch --output-format csv -- $SWEEP > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_sr25_write -o ${tag}_sr25_write --output-format csv -- $SWEEP > /dev/null 2>&1
python3 tools/bench_configs.py > $out/${tag}_other_configs.jsonl 2> /dev/null
python3 tools/bench_mnist.py 1024 > $out/${tag}_mnist_1024.json 2> /dev/null
python3 tools/bench_mnist.py 4096 > $out/${tag}_mnist_4096.json 2> /dev/null
