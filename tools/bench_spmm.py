"""HBM rate of the stand-alone multi-support SpMM (gml_spmm_fwd: H[r, s, :] = sum_k val[k, s] x[col[k], :], H materialised)
beside the fused forward that never writes H.  python tools/bench_spmm.py   (GML_SPMM_BATCH=<graphs>, default 32768)"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gnn_matlang_amd import functional as Fn, _lib

dev = torch.device('cuda:0')
data, _ = bench.build_batch(int(os.environ.get('GML_SPMM_BATCH', '32768')), 2048, seed=1000, device=dev)
csr = data.csr('edge_index2')
N, E, S, Fin, Fout = csr.N, csr.E, 8, 32, 30
val = csr.sort_values(data.edge_attr2)
x = torch.randn(N, Fin, device=dev)
w = torch.randn(S, Fin, Fout, device=dev) * 0.1


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
    return a.elapsed_time(b) / n * 1e-3


h = torch.empty(N, S * Fin, device=dev)
L = _lib.lib()


def spmm(records=True):
    _lib.call('gml_spmm_fwd', Fn._ptr(csr.rowptr), Fn._ptr(csr.col), Fn._ptr(csr.ginfo128 if records else None), Fn._ptr(None),
              Fn._ptr(val), Fn._ptr(x), Fin, Fn._ptr(h), N, S, Fin, Fn._stream(dev))


out = torch.empty(N, Fout, device=dev)
gi, gflag = Fn.fwd_groups(csr, x, S, Fin, Fout)


def fused():
    Fn._fused_conv(csr.rowptr, csr.col, gi, None, val, x, Fin, w, (Fin * Fout, Fout, 1), None, out, Fout, N, S, Fin, Fout, gflag, 0)


t_spmm, t_fused = timeit(spmm), timeit(fused)
h1 = h.clone()
t_spmm_rows = timeit(lambda: spmm(False))
assert os.environ.get('GML_SKIP_CHECK') or torch.allclose(h, h1, rtol=1e-5, atol=1e-5)
q_spmm = 4 * (E * S + N * Fin + N * S * Fin) + 4 * (E + N + 1)
q_fused, _ = Fn.conv_cost(N, E, S, Fin, Fout)
print(json.dumps(dict(N=N, E=E, S=S, Fin=Fin, Fout=Fout,
                      spmm_row_per_lane_group_us=t_spmm_rows * 1e6,
                      spmm=dict(us=t_spmm * 1e6, algorithmic_MB=q_spmm / 1e6, GBps=q_spmm / t_spmm / 1e9, frac_of_8TBps=q_spmm / t_spmm / 8e12),
                      fused_conv=dict(us=t_fused * 1e6, algorithmic_MB=q_fused / 1e6, GBps=q_fused / t_fused / 1e9,
                                      frac_of_8TBps=q_fused / t_fused / 8e12,
                                      time_vs_spmm_hbm_floor=t_fused / (q_spmm / 8e12)))))
