"""Per-phase cycle shares of gml_k_spectconv_fwd2 (library built with -DGML_FWD2_TIMING: tools/build_variant.py fwtiming -DGML_FWD2_TIMING)."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gnn_matlang_amd import _lib, models

dev = torch.device('cuda:0')
data, _ = bench.build_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 131072, 2048, seed=1000, device=dev)
data.csr('edge_index2')
torch.manual_seed(0)
model = models.zinc_gnnml3().to(dev)
L = _lib.lib()
L.gml_debug_fwd2_prof.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 32)()
for it in range(3):
    loss = models.zinc_loss(model(data), data.y)
    loss.backward()
    torch.cuda.synchronize()
    L.gml_debug_fwd2_prof(buf, 1)
names = ['commit (waits for the prefetched registers)', 'barrier', 'issue next group (record, then loads)', 'own-row loads + row bounds',
         'aggregation', 'value gather issue', 'projection', 'output stores', 'Hadamard branch', 'end barrier']
names3 = ['LOADER: wait for the landing DMAs (vmcnt)', 'barrier (compute waves)', 'LOADER: issue next group (records from LDS, DMA)', 'row bounds',
          'aggregation (+ own x row)', 'LOADER: barrier', 'projection', 'output stores', 'Hadamard branch', '-']
for title, nm, b in (('fwd2 (register staging)', names, buf[:16]), ('fwd3 (LDS-DMA ring)', names3, buf[16:])):
    tot = float(sum(b))
    if not tot:
        continue
    print('# ' + title)
    for n, v in zip(nm, b):
        print('%-48s %14d  %5.1f%%' % (n, v, 100.0 * v / tot if tot else 0))
    print('cycles per step per wave (all launches of this kernel): %.0f' % (tot / 256 / 8))
