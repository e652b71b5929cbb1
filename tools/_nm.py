import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import torch
from gnn_matlang_amd import _lib
from gnn_matlang_amd.graph import _ptr, _stream
dev = torch.device('cuda:0')
N = 1502656
def run(Fin, F2, ldo, ldx=None):
    ldx = ldx or Fin
    x = torch.randn(N, ldx, device=dev); out = torch.zeros(N, ldo, device=dev)
    w11 = torch.randn(F2, Fin, device=dev); w12 = torch.randn(F2, Fin, device=dev); b = torch.randn(F2, device=dev)
    f = lambda: _lib.call('gml_node_mix_fwd', _ptr(x), ldx, _ptr(w11), _ptr(b), _ptr(w12), _ptr(b), _ptr(out), ldo, N, Fin, F2, _stream(dev))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print('Fin %d F2 %d ldo %d ldx %d: %.3f ms' % (Fin, F2, ldo, ldx, e0.elapsed_time(e1) / 10))
run(32, 2, 32); run(32, 2, 2); run(32, 16, 48); run(32, 16, 16); run(25, 2, 32); run(8, 2, 32)
