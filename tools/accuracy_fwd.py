#!/usr/bin/env python3
"""Error of the fused forward vs an fp64 reference, for both projection arithmetics."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_batch
from gnn_matlang_amd import functional as Fn, _lib

dev = torch.device('cuda:0')
data, _ = build_batch(4096, 2048, 1000, dev)
csr = data.csr()
N, E = csr.N, csr.E
val = csr.sort_values(data.edge_attr2)
torch.manual_seed(0)
for (Fin, Fout, scale) in ((32, 30, 1.0), (25, 30, 1.0), (32, 30, 100.0)):
    x = torch.randn(N, Fin, device=dev) * scale
    w = torch.randn(8, Fin, Fout, device=dev) * 0.2
    b = torch.randn(Fout, device=dev)
    # fp64 reference: H = A^T X per support, then projection
    h = Fn.spmm(csr, val, x.contiguous(), 8, Fin).double()
    ref = h @ w.double().view(8 * Fin, Fout) + b.double()
    for name, fl in (('bf16x3', 0), ('f32 mfma', _lib.GML_F32_MFMA)):
        out = torch.empty(N, Fout, device=dev)
        Fn._fused_conv(csr.rowptr, csr.col, csr.ginfo, None, val, x, Fin, w, (Fin * Fout, Fout, 1), b, out, Fout, N, 8, Fin, Fout, fl, 0)
        err = (out.double() - ref).abs()
        print('Fin %d scale %g %-9s max|err|/max|ref| = %.2e   max elementwise rel (|ref|>1e-3 max) = %.2e' % (
            Fin, scale, name, (err.max() / ref.abs().max()).item(),
            (err / ref.abs().clamp(min=1e-3 * ref.abs().max())).max().item()))
