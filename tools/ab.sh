#!/bin/bash
# A/B of builds of libgml_hip.so on ONE box, interleaved: tools/ab.sh "<v1> <v2> ..." [reps] [bench args]
# (variants = _ab/lib_<v>.so from tools/build_variant.py; 'tree' = the in-tree library)
vs=$1; reps=${2:-2}; shift; shift
for rep in $(seq 1 $reps); do
  for v in $vs; do
    if [ $v = tree ]; then lib=$PWD/gnn_matlang_amd/libgml_hip.so; else lib=$PWD/_ab/lib_$v.so; fi
    GML_LIB=$lib python3 bench.py --no-cpu --ref-batch 0 --no-extras --min-seconds 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d.get('kernels_ms_per_step'))"
  done
done
