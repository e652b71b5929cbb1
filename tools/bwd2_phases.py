"""Per-phase cycle shares of gml_k_spectconv_bwd2 (library built with GML_CXXFLAGS=-DGML_BWD2_TIMING)."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gnn_matlang_amd import _lib, functional as Fn, models

dev = torch.device('cuda:0')
data, _ = bench.build_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 2048, seed=1000, device=dev)
csr = data.csr('edge_index2')
torch.manual_seed(0)
model = models.zinc_gnnml3().to(dev)
L = _lib.lib()
L.gml_debug_bwd2_prof.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 16)()
for it in range(3):
    loss = models.zinc_loss(model(data), data.y)
    loss.backward()
    torch.cuda.synchronize()
    L.gml_debug_bwd2_prof(buf, 1)
names = ['top barrier (bwd2: + dW of the previous group)', 'commit staged regs + barrier', 'Z projection', 'edge phase', 'barrier', 'dX chain + dx stores', 'tail', 'dW contraction (bwd3)',
         'next loads issued', 'dval stores', 'P split', 'dW images + slab barriers (bwd3)', '-', '-', '-', '-']
tot = float(sum(buf))
for n, v in zip(names, buf):
    if v:
        print('%-52s %12d  %5.1f%%' % (n, v, 100.0 * v / tot if tot else 0))
print('cycles per launch per workgroup: %.0f' % (tot / 4 / 256))
