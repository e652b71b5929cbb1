python3 -m pytest tests -m gpu -x -q > gpurun_out/r02c_pytest.log 2>&1; head -60 gpurun_out/r02c_pytest.log; echo ...; grep -n "Fatal\|passed\|failed\|rror" gpurun_out/r02c_pytest.log | head
