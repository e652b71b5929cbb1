export TMPDIR=/tmp
python3 tools/bench_index_build.py 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_idx -o idx -- python3 tools/bench_index_build.py > gpurun_out/r02_idx.log 2>&1
python3 tools/rocprof_summary.py gpurun_out/r02_idx/idx_results.db | head -30
