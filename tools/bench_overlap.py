"""Do the edge-branch kernels overlap with the node-branch kernels when they run on a second stream?
(ZINC GNNML3 layer at the bench size; sequential vs two-stream timing of  edge_fwd || conv_fwd  and  edge_bwd || conv_bwd.)
python tools/bench_overlap.py [graphs]"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gnn_matlang_amd import functional as Fn, models
from gnn_matlang_amd.spect_conv import _sorted_values

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = torch.device('cuda:0')
data, _ = bench.build_batch(B, 2048, seed=1000, device=dev)
csr = data.csr('edge_index2')
torch.manual_seed(0)
m = models.zinc_gnnml3().to(dev)
L = m.conv2
val = _sorted_values(csr, data.edge_index2, data.edge_attr2, None)
N = csr.N
x = torch.randn(N, L.conv1.weight.size(1), device=dev)
w = [L.fc1_1.weight.detach(), L.fc1_2.weight.detach(), L.fc1_3.weight.detach(), L.fc1_4.weight.detach()]
cw = L.conv1.weight.detach()
S, Fin, Fout = cw.shape
ea, ea_t = Fn.edge_mlp_fwd(val, *w, csr.tpos, csr.presplit(val))
G = torch.randn(N, Fout, device=dev)
dea = torch.randn_like(ea)
val_s = csr.to_source_order(val)
pre_s = csr.presplit(val_s)
side = torch.cuda.Stream()


def edge_fwd():
    return Fn.edge_mlp_fwd(val, *w, csr.tpos, csr.presplit(val))


def conv_fwd():
    return Fn.ML3LayerFunction.apply(x, ea, None, None, None, None, cw, L.conv1.bias.detach(), L.fc11.weight.detach(), L.fc11.bias.detach(),
                                     L.fc12.weight.detach(), L.fc12.bias.detach(), csr, False, L.nout2)


def edge_bwd():
    return Fn.edge_mlp_bwd(val_s, *w, dea, False, pre_s)


def conv_bwd():
    return Fn.fused_conv_bwd(csr, ea_t, x, G, cw, True, True, True)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def both(a, b):
    def f():
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            a()
            done = torch.cuda.Event()
            done.record()
        b()
        torch.cuda.current_stream().wait_event(done)
    return f


out = dict(graphs=B, edges=int(csr.E))
for name, a, b in (('fwd', edge_fwd, conv_fwd), ('bwd', edge_bwd, conv_bwd)):
    ta, tb = timeit(a), timeit(b)
    tseq = timeit(lambda: (a(), b()))
    tpar = timeit(both(a, b))
    out[name] = dict(edge_ms=ta, conv_ms=tb, sequential_ms=tseq, two_streams_ms=tpar)
print(json.dumps(out))
