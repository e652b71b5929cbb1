export TMPDIR=/tmp
for r in 1 2; do
python3 bench.py --no-cpu --ref-batch 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd2', round(d['ms_per_step'],4), d.get('kernels_ms_per_step'))"
GML_FWD64=1 python3 bench.py --no-cpu --ref-batch 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd64', round(d['ms_per_step'],4), d.get('kernels_ms_per_step'))"
done
BENCH="python3 bench.py --steps 4 --warmup 2 --no-cpu --no-profile --ref-batch 0"
export GML_FWD64=1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/f64_pmcA -o f64_pmcA --output-format csv -- $BENCH > gpurun_out/f64_pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES -d gpurun_out/f64_pmcB -o f64_pmcB --output-format csv -- $BENCH > gpurun_out/f64_pmcB.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES -d gpurun_out/f64_pmcC -o f64_pmcC --output-format csv -- $BENCH > gpurun_out/f64_pmcC.log 2>&1
unset GML_FWD64
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES -d gpurun_out/r02b_pmcC -o r02b_pmcC --output-format csv -- $BENCH > gpurun_out/r02b_pmcC.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/r02b_pmcA -o r02b_pmcA --output-format csv -- $BENCH > gpurun_out/r02b_pmcA.log 2>&1
