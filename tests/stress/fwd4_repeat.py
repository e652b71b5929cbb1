"""Fresh-process repeat-and-compare of one fused forward launch (the reproducer of profiles/r04_fwd4_nondeterminism.txt; not
collected by pytest -- run it through tests/stress/run_fwd4.sh on a GPU box).  Env: DS supports, DF input features, DEG entries per row,
DN rows, REPS launches.  Every launch is compared with the CPU oracle; the last line says how many differed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnn_matlang_amd import SpectConv
from oracle import spect_conv_oracle as O
dev = torch.device('cuda:0')
S, fin, fout = int(os.environ.get('DS', 6)), int(os.environ.get('DF', 48)), 32
deg, spread, N = int(os.environ.get('DEG', 13)), 12, int(os.environ.get('DN', 700))
rng = np.random.default_rng(1)
src = np.repeat(np.arange(N), deg)
dst = np.clip(src + rng.integers(-spread, spread + 1, size=src.shape), 0, N - 1)
ei = np.unique(np.vstack((src, dst)), axis=1).astype(np.int64)
T = torch.tensor
E = ei.shape[1]
torch.manual_seed(0)
ea, x = torch.randn(E, S), torch.randn(N, fin)
m = SpectConv(fin, fout, S, selfconn=False).to(dev)
w0 = m.weight.detach().cpu()
yo = O.spectconv_forward(x, T(ei), ea, w0, torch.zeros(fout), False)
xd, ed, eid = x.to(dev), ea.to(dev), T(ei).to(dev)
junk = torch.randn(64, 1024, 1024, device=dev)
nbad = 0
for rep in range(int(os.environ.get('REPS', 12))):
    with torch.no_grad():
        m.bias.zero_()
        if rep % 2: junk.mul_(1.0001)            # other kernels in between
        y = m(xd, eid, ed).cpu()
    err = (y - yo).abs().max(1).values
    nan = int(torch.isnan(y).any(1).sum())
    bad = (~(err <= 1e-3 * float(yo.abs().max()))).nonzero().flatten().tolist()
    nbad += len(bad) > 0
    if os.environ.get('CNT'):
        import ctypes
        from gnn_matlang_amd import _lib
        arr = (ctypes.c_ulonglong * 8)()
        _lib.lib().gml_debug_f4_counts(arr, 1)
        print('   lds-vs-global mismatches VAL/COL/X/RP, checks:', list(arr)[:7])
    print('rep', rep, 'nan rows', nan, 'bad rows', len(bad), bad[:24])
print(os.environ.get('GML_LIB', 'default'), 'S', S, 'Fin', fin, 'N', N, 'failing reps', nbad)
