#!/bin/bash
# A/B of two builds of libgml_hip.so on the SAME box: tools/ab_bench.sh <dir with lib_a.so lib_b.so> [bench args]
d=$1; shift
for rep in 1 2; do
  for v in a b; do
    cp $d/lib_$v.so gnn_matlang_amd/libgml_hip.so
    python bench.py --no-cpu --ref-batch 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['kernels_ms_per_step'])"
  done
done
