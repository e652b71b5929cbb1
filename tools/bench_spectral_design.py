"""Support precompute: device kernels (gml_spectral_count + gml_spectral_design) vs the batched host (numpy/LAPACK)
implementation, graphs per second.  python tools/bench_spectral_design.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_matlang_amd import SpectralDesign, synthetic

dev = torch.device('cuda:0')
res = []
for name, kind, count, kw in (('ZINC-12k (recfield 2, nfreq 7, dv 2)', 'zinc', 12000, dict(recfield=2, dv=2, nfreq=7)),
                              ('MNIST-75 (recfield 3, nfreq 5, dv 10)', 'mnist75', 4000, dict(recfield=3, dv=10, nfreq=5))):
    raw = synthetic.make_graphs(kind, count, seed=1)
    sd = SpectralDesign(**kw)
    sizes = np.array([np.asarray(x).shape[0] for x, _, _ in raw])
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    X = torch.from_numpy(np.concatenate([np.asarray(x, np.float32)[:, :1] for x, _, _ in raw])).to(dev)
    EI = torch.from_numpy(np.concatenate([np.asarray(ei, np.int64) + ptr[i] for i, (_, ei, _) in enumerate(raw)], 1)).to(dev)
    P = torch.tensor(ptr, dtype=torch.int32, device=dev)
    d = sd.design_device(X, EI, P)                                     # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        d = sd.design_device(X, EI, P)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / reps
    nh = min(count, 2000)
    t0 = time.perf_counter()
    host = sd.design_many(raw[:nh])
    t_host = time.perf_counter() - t0
    m_host = sum(h['edge_index2'].shape[1] for h in host)
    m_dev = int((d['edge_index2'][0] < int(ptr[nh])).sum().item())
    res.append(dict(workload=name, graphs=count, mask_entries=int(d['edge_index2'].size(1)),
                    device_ms=t_dev * 1e3, device_graphs_per_s=count / t_dev,
                    host_graphs_per_s=nh / t_host, host_sample=nh, same_mask_size_on_sample=bool(m_host == m_dev)))
    print(json.dumps(res[-1]))
