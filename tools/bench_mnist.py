"""MNIST-75 GNNML3 (config 4) train step: sparse (block-CSR HIP kernels) vs dense-block with torch.bmm ('dense_lib', the
round-1 path) vs dense-block with the HIP batched support product ('dense', csrc/gml_dense.hip).
python tools/bench_mnist.py [graphs]"""
import json
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_matlang_amd import SpectralDesign, collate, models, synthetic, dense_block

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ONLY = sys.argv[2] if len(sys.argv) > 2 else None          # e.g. `dense`: profile one path alone
dev = torch.device('cuda:0')
raw = synthetic.make_graphs('mnist75', 64, seed=1)
pool = SpectralDesign(recfield=3, dv=10, nfreq=5).design_many(raw)
data = collate([pool[i % 64] for i in range(B)]).to(dev)
data.y = torch.randint(0, 10, (B,), device=dev)
out = {}
for name, dn in (('sparse', 0), ('dense_lib', 75), ('dense', 75)):
    if ONLY and name != ONLY:
        continue
    dense_block.USE_LIBRARY = name == 'dense_lib'
    torch.manual_seed(0)
    m = models.mnist_gnnml3(dense_n=dn).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)

    def step():
        opt.zero_grad(set_to_none=True)
        l = models.mnist_loss(m(data), data.y)
        l.backward()
        opt.step()
        return l
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        l = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out[name] = dict(ms_per_step=dt * 1e3, graphs_per_s=B / dt, loss=float(l))
    if name == 'dense':
        # roofline of the dominant hand-written kernel: live HIP events around every tagged launch, a second pass
        from gnn_matlang_amd import functional as Fn
        Fn.PROFILE = {}
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        summ = Fn.profile_summary(Fn.PROFILE)
        Fn.PROFILE = None
        BF16_PEAK = 2500.0                                        # dense bf16 MFMA TFLOP/s (MI355X_MICROARCH.md); bf16x3 = 3 products per fp32 product
        kern = {k: dict(launches_per_step=v['launches'] / n, avg_launch_ms=round(v['ms'], 4), ms_per_step=round(v['ms'] * v['launches'] / n, 4),
                        GBps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1), GBps_compulsory=round(v['bytes_compulsory'] / (v['ms'] * 1e-3) / 1e9, 1),
                        TFLOPs=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2))
                for k, v in summ.items() if v['bytes'] > 0}
        top = max((k for k in kern if k.startswith('dense_conv')), key=lambda k: kern[k]['ms_per_step'], default=None)
        if top is not None:
            # priced against the COMPULSORY bytes (no Hcat -- an array this design writes for its own weight gradient) and
            # against the bf16 matrix roof at three products per fp32 product; whichever takes longer binds
            v = summ[top]
            t = v['ms'] * 1e-3
            t_hbm, t_mat = v['bytes_compulsory'] / 8000e9, 3 * v['flops'] / (BF16_PEAK * 1e12)
            if t_mat > t_hbm:
                roof = dict(bound='mfma', achieved=3 * v['flops'] / t / 1e12, peak=BF16_PEAK, unit='TFLOP/s', frac=t_mat / t)
            else:
                roof = dict(bound='hbm', achieved=v['bytes_compulsory'] / t / 1e9, peak=8000.0, unit='GB/s', frac=t_hbm / t)
            roof.update(kernel=top, traffic=None, avg_launch_ms=v['ms'], algorithmic_bytes_per_launch=v['bytes_compulsory'],
                        bytes_per_launch_with_Hcat=v['bytes'], frac_with_Hcat_bytes=v['bytes'] / t / 1e9 / 8000.0,
                        frac_of_bf16_matrix_roof=t_mat / t,
                        note='mean over the three layers of the model; compulsory bytes = packed support images + x + out (Hcat, written for this '
                             'design\'s own dW, is NOT counted; with it: frac_with_Hcat_bytes); 3 bf16 products per fp32 product on the matrix pipe')
            out[name]['roofline'] = roof
        out[name]['kernels'] = kern
print(json.dumps(dict(graphs=B, nodes=int(data.x.size(0)), support_edges=int(data.edge_index2.size(1)), **out)))
