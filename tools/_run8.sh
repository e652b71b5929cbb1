for r in 1 2; do
for cfg in "2 8" "3 8" "3 4"; do set -- $cfg
GML_BWD_LAYOUT=$1 GML_BWD_NW=$2 python3 bench.py --no-cpu --ref-batch 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('layout$1 nw$2', round(d['ms_per_step'],4), d.get('kernels_ms_per_step'))"
done; done
for cfg in "3 8" "3 4"; do set -- $cfg
echo "== layout $1 nw $2"
GML_BWD_LAYOUT=$1 GML_BWD_NW=$2 GML_LIB=$PWD/_ab/lib_timing.so python3 tools/bwd2_phases.py 2>&1 | grep -v "amdgpu.ids"
done
GML_BWD_NW=8 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
