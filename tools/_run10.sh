python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8
python3 bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.log; tail -20 gpurun_out/r02c_bench.log; python3 -c "
import json; d=json.load(open('gpurun_out/r02c_bench.json'))
for k in ('value','ms_per_step','blocks','value_exact_fp32','fresh_batch','ref_batch','epoch_bs64','kernels_ms_per_step'): print(k, d.get(k))
print('roofline', {k:d['roofline'][k] for k in ('frac','avg_launch_ms','achieved')})
print('cpu', {k:v for k,v in d['cpu_baseline'].items() if k in ('value','cores','cpu_model','host_cores','bs64','thread_ladder')})
"
