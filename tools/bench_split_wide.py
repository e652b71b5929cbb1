"""Micro-benchmark of gml_ml3_split_bwd at the wide Hadamard shapes (counting / sr25 / mutag layers): python tools/bench_split_wide.py [N]
(GML_SPLIT_MM=0 in the environment: the row-per-lane form)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_matlang_amd import functional as Fn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
dev = torch.device('cuda:0')
torch.manual_seed(0)


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


for name, Fin, nout1, F2 in (('counting', 32, 16, 16), ('sr25', 48, 32, 16), ('mutag', 48, 24, 24), ('64/24', 64, 24, 24), ('16/8', 16, 8, 8)):
    C = nout1 + F2
    gy, y = torch.randn(N, C, device=dev), torch.randn(N, C, device=dev)
    x = torch.randn(N, Fin, device=dev)
    w = [torch.randn(F2, Fin, device=dev) * 0.2, torch.randn(F2, device=dev), torch.randn(F2, Fin, device=dev) * 0.2, torch.randn(F2, device=dev)]
    fn = lambda: Fn.ml3_split_bwd(gy, y, nout1, x, *w, need_dx=True, need_dcb=True)
    mb = 4 * N * (2 * C + nout1 + nout1 + 2 * Fin) / 1e6          # gy, y (conv columns), G out, x in, dx out
    us = timeit(fn)
    print('%s Fin=%d nout1=%d F2=%d: %.1f us, %.0f MB -> %.2f TB/s' % (name, Fin, nout1, F2, us, mb, mb / us))
