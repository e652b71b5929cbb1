cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
GML_LIB=$PWD/_ab/lib_fwtiming.so timeout 300 python tools/fwd2_phases.py 2>&1 | tail -13
for v in 0 1 0 1; do
GML_FWD_DMA=$v timeout 300 python bench.py --no-cpu --ref-batch 0 --no-extras --min-seconds 0 --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('DMA=$v', round(d['ms_per_step'],4), d.get('kernels_ms_per_step'))"
done
