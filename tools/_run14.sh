python3 -m pytest tests -m gpu -q 2>&1 | tail -4
python3 tools/bench_index_build.py 2>&1 | grep -v amdgpu
python3 bench.py --no-cpu 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read())
for k in ('value','ms_per_step','fresh_batch','epoch_bs64','ref_batch'): print(k, d.get(k))"
