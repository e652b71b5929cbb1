"""Micro-benchmark of the edge-branch kernels (S = 8, ZINC batch size) : python tools/bench_edge_mlp.py [E]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_matlang_amd import functional as Fn

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4580224
S = 8
dev = torch.device('cuda:0')
torch.manual_seed(0)
ea = torch.randn(E, S, device=dev)
ws = [torch.randn(2 * S, S, device=dev) * 0.7 for _ in range(3)] + [torch.randn(S, 4 * S, device=dev) * 0.5]
gout = torch.randn(E, S, device=dev)
tp = torch.randperm(E, device=dev).int()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print('mode', 'VALU' if os.environ.get('GML_EDGE_VALU') == '1' else 'chain', 'E', E)
print('fwd          %.1f us' % timeit(lambda: Fn.edge_mlp_fwd(ea, *ws)))
print('fwd dual     %.1f us' % timeit(lambda: Fn.edge_mlp_fwd(ea, *ws, tpos=tp)))
print('bwd (no gin) %.1f us' % timeit(lambda: Fn.edge_mlp_bwd(ea, *ws, gout, False)))
print('bwd (gin)    %.1f us' % timeit(lambda: Fn.edge_mlp_bwd(ea, *ws, gout, True)))
