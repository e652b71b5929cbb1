#!/bin/bash
# Census of packed-FP32 operand-select forms in the device code of every kernel (DESIGN s4.1c: v_pk_fma_f32 ... op_sel:[0,1,0]
# -- the low product reading the HIGH half of src1 -- lost results on gfx950 in gml_k_spectconv_fwd4's divergent loop).
#   tools/opsel_census.sh > profiles/rNN_opsel_census.txt
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
ls $root/gnn_matlang_amd/csrc/*.hip | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$root/include -I$root/gnn_matlang_amd/csrc -S --cuda-device-only -o $tmp/\$(basename {} .hip).s {} 2>/dev/null"
echo "# forms over the whole library (instruction, operand-select suffix, count)"
grep -h "v_pk_[a-z_0-9]* .*op_sel" $tmp/*.s | awk '{print $1, $NF}' | sort | uniq -c | sort -rn
echo
echo "# kernels that contain v_pk_fma_f32 ... op_sel:[0,1,0] (count, kernel)"
for f in $tmp/*.s; do awk -v F=$(basename $f .s) '/^_Z.*:/{name=$1} /v_pk_fma_f32.*op_sel:\[0,1,0\]/{c[name]++} END{for(n in c) print c[n], F, n}' $f; done | sort -rn | c++filt | cut -c1-200
rm -rf $tmp
