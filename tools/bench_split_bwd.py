"""Micro-benchmark of gml_ml3_split_bwd at the ZINC batch shape: python tools/bench_split_bwd.py [N]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_matlang_amd import functional as Fn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 751328
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


for Fin, nout1, F2, need_dx in ((32, 30, 2, True), (25, 30, 2, False), (32, 30, 2, False), (0, 32, 0, False)):
    C = nout1 + F2
    gy, y = torch.randn(N, C, device=dev), torch.randn(N, C, device=dev)
    if F2:
        x = torch.randn(N, Fin, device=dev)
        w = [torch.randn(F2, Fin, device=dev), torch.randn(F2, device=dev), torch.randn(F2, Fin, device=dev),
             torch.randn(F2, device=dev)]
        fn = lambda: Fn.ml3_split_bwd(gy, y, nout1, x, *w, need_dx=need_dx, need_dcb=True)
        mb = 4 * N * (3 * C + Fin * (2 if need_dx else 1)) / 1e6
    else:
        fn = lambda: Fn.ml3_split_bwd(gy, y, nout1, need_dcb=True)
        mb = 4 * N * 3 * C / 1e6
    us = timeit(fn)
    print('Fin=%d nout1=%d F2=%d dx=%s: %.1f us, %.0f MB -> %.2f TB/s' % (Fin, nout1, F2, need_dx, us, mb, mb / us))
