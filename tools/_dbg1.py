import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gnn_matlang_amd import functional as Fn
from gnn_matlang_amd.graph import GraphCSR
torch.manual_seed(0)
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
N = 203
src = np.repeat(np.arange(N), 5); dst = np.clip(src + rng.integers(-6, 7, size=src.shape), 0, N-1)
ei = np.unique(np.vstack((src, dst)), axis=1)
csr = GraphCSR.from_edge_index(torch.tensor(ei).to(dev), N)
E = ei.shape[1]
for (S, Fin, Fout) in [(6,64,2),(2,64,2),(6,96,2),(6,64,2)]:
    val = torch.randn(E, S, device=dev); x = torch.randn(N, Fin, device=dev); w = torch.randn(S, Fin, Fout, device=dev)
    out = torch.empty(N, Fout, device=dev)
    Fn._fused_conv(csr.rowptr, csr.col, csr.ginfo, None, val, x, Fin, w, (Fin*Fout, Fout, 1), None, out, Fout, N, S, Fin, Fout, 0, 0)
    h = Fn.spmm(csr, val, x, S, Fin)
    ref = h @ w.view(S*Fin, Fout)
    err = ((out-ref).abs().max()/ref.abs().max()).item()
    # per-column error
    bad = ((out-ref).abs().max(1).values > 1e-3*ref.abs().max()).nonzero().flatten().tolist()
    print(bad)
    print(csr.ginfo.tolist())
    print(S, Fin, Fout, 'err %.2e' % err, 'rows bad', int(((out-ref).abs().max(1).values > 1e-3*ref.abs().max()).sum()))
