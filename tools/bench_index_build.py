"""Per-batch index work of a fresh batch (what `fresh_batch` of bench.py adds to a step): CSR of both views, group
records, target-order / source-order value copies, bf16 pre-splits.  python tools/bench_index_build.py [graphs]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gnn_matlang_amd.graph import Batch

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
data, _ = bench.build_batch(n, 2048, seed=1000, device=dev)
fields = {k: v for k, v in data.__dict__.items() if not k.startswith('_')}


def build():
    b = Batch(**fields)
    csr = b.csr('edge_index2')
    val = csr.sort_values(b.edge_attr2)
    csr.presplit(val)
    vt = csr.to_source_order(val, cache=True)
    csr.presplit(vt)
    return csr


for _ in range(3):
    build()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    build()
torch.cuda.synchronize()
print('index build: %.3f ms per batch of %d graphs (%d support edges)' % ((time.perf_counter() - t0) / reps * 1e3, n, data.edge_index2.size(1)))
