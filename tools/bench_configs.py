"""Train-step time of the other BASELINE configs (parity-test cases, not bench lines) on large synthetic batches, with the
per-tag kernel breakdown of functional.PROFILE: shows which configs run on fused kernels and which on fallbacks.
python tools/bench_configs.py"""
import json
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_matlang_amd import SpectralDesign, collate, models, synthetic, functional as Fn

dev = torch.device('cuda:0')
Fn.VERBOSE = True            # record the kernel family of every layer call (functional.PATHS)
CFG = [('counting', 'counting', 256, 64, dict(recfield=1, dv=1, nfreq=10, addadj=True), lambda: models.counting_gnnml3(1, 12), models.counting_loss, 1),
       ('sr25', 'regular', 64, 32, dict(recfield=1, dv=1, nfreq=5), lambda: models.sr25_gnnml3(1, 6), None, 1),
       ('mutag_gnnml3', 'zinc', 512, 32, dict(recfield=1, dv=1, nfreq=3), lambda: models.mutag_gnnml3(21, 4), models.mutag_loss, 21)]
for name, kind, pool_n, reps, kw, ctor, loss, fdim in CFG:
    raw = synthetic.make_graphs(kind, pool_n, seed=2)
    pool = SpectralDesign(**kw).design_many(raw)
    host = collate(pool * reps)
    if host.x.shape[1] != fdim:
        host.x = host.x[:, :fdim].contiguous() if host.x.shape[1] > fdim else torch.ones(host.x.shape[0], fdim)
    data = host.to(dev)
    B = data.num_graphs
    data.y = torch.rand(B, device=dev)
    torch.manual_seed(0)
    m = ctor().to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        pre = m(data)
        l = loss(pre, data.y) if loss is not None else pre.square().sum()
        l.backward()
        opt.step()
    Fn.PATHS.clear()
    step()
    paths = dict(Fn.PATHS)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    Fn.PROFILE = {}
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    summ = Fn.profile_summary(Fn.PROFILE)
    Fn.PROFILE = None
    print(json.dumps(dict(config=name, graphs=B, nodes=int(data.x.size(0)), support_edges=int(data.edge_index2.size(1)),
                          S=int(data.edge_attr2.size(1)), ms_per_step=round(dt * 1e3, 3), graphs_per_s=round(B / dt),
                          kernels_ms_per_step={k: round(v['ms'] * v['launches'] / n, 3) for k, v in summ.items()},
                          kernel_paths_per_step=paths,
                          slow_paths=sorted(k for k in paths if 'UNFUSED' in k or 'library GEMMs' in k or 'VALU kernels' in k))))
