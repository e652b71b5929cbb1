export TMPDIR=/tmp
for cfg in "2 8" "3 8" "3 4"; do set -- $cfg
echo "== layout $1 nw $2"
GML_BWD_LAYOUT=$1 GML_BWD_NW=$2 GML_LIB=$PWD/_ab/lib_timing.so python3 tools/bwd2_phases.py 2>&1 | grep -v "^-  \|amdgpu.ids"
done
BENCH="python3 bench.py --steps 4 --warmup 2 --no-cpu --no-profile --ref-batch 0"
for cfg in "3 8" "3 4"; do set -- $cfg
export GML_BWD_LAYOUT=$1 GML_BWD_NW=$2
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES -d gpurun_out/b3_$2_pmcC -o pmcC --output-format csv -- $BENCH > gpurun_out/b3_pmcC.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/b3_$2_pmcA -o pmcA --output-format csv -- $BENCH > gpurun_out/b3_pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES -d gpurun_out/b3_$2_pmcB -o pmcB --output-format csv -- $BENCH > gpurun_out/b3_pmcB.log 2>&1
done
