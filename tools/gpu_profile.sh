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
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o ${tag}_stats -- python3 bench.py --steps 20 --warmup 5 --batch $batch --no-cpu --no-profile --ref-batch 0 --min-seconds 0 > $out/${tag}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/${tag}_pmcA -o ${tag}_pmcA --output-format csv -- $BENCH > $out/${tag}_pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES -d $out/${tag}_pmcB -o ${tag}_pmcB --output-format csv -- $BENCH > $out/${tag}_pmcB.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES -d $out/${tag}_pmcC -o ${tag}_pmcC --output-format csv -- $BENCH > $out/${tag}_pmcC.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_fetch -o ${tag}_fetch --output-format csv -- $BENCH > $out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_write -o ${tag}_write --output-format csv -- $BENCH > $out/${tag}_write.log 2>&1
find $out/${tag}_* -name '*.csv' | head -40
