#!/bin/bash
# same-box comparison of builds of libgml_hip.so for the stand-alone SpMM: tools/ab_spmm.sh <dir> <variant letters...>
d=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    cp $d/lib_$v.so gnn_matlang_amd/libgml_hip.so
    python tools/bench_spmm.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v spmm %.1f us %.3f of roof; fused conv %.1f us' % (d['spmm']['us'], d['spmm']['frac_of_8TBps'], d['fused_conv']['us']))"
  done
done
