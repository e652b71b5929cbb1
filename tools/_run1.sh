python3 -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest.log 2>&1; tail -3 gpurun_out/r02a_pytest.log
bash tools/gpu_profile.sh r02a
GML_LIB=$PWD/_ab/lib_timing.so python3 tools/bwd2_phases.py > gpurun_out/r02a_phases.log 2>&1; cat gpurun_out/r02a_phases.log
