export GML_BENCH_NOCHECK=1 GML_BWD_LAYOUT=3 GML_BWD_NW=8
for v in tree abl1 abl2 abl4 abl8 abl16 abl32 abl63 tree; do
if [ $v = tree ]; then lib=$PWD/gnn_matlang_amd/libgml_hip.so; else lib=$PWD/_ab/lib_$v.so; fi
GML_LIB=$lib python3 bench.py --no-cpu --ref-batch 0 --steps 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().replace('NaN','null')); print('$v', round(d['ms_per_step'],4), d.get('kernels_ms_per_step'))"
done
export GML_BWD_NW=4
for v in tree abl1 abl2 abl4 abl63; do
if [ $v = tree ]; then lib=$PWD/gnn_matlang_amd/libgml_hip.so; else lib=$PWD/_ab/lib_$v.so; fi
GML_LIB=$lib python3 bench.py --no-cpu --ref-batch 0 --steps 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().replace('NaN','null')); print('nw4 $v', round(d['ms_per_step'],4), d.get('kernels_ms_per_step'))"
done
