"""Train-step time of the other BASELINE configs (parity-test cases, not bench lines) on large synthetic batches, with the
per-tag kernel breakdown of functional.PROFILE, the kernel family every layer call took, and -- per config -- the same
`roofline` record bench.py carries for the ZINC step: dominant kernel, SURVEY s8(d) algorithmic bytes per launch, live mean
launch time (HIP events), fraction of the HBM roof.

    python tools/bench_configs.py [--quick]          one JSON line per config

bench.py embeds `run(dev, quick=True)` as its `other_configs` block (a few seconds)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def _configs():
    from gnn_matlang_amd import models
    # (name, synthetic kind, pool graphs, repeats (full, quick), SpectralDesign kwargs, model, loss, input features, reference)
    return [
        ('counting', 'counting', 256, (64, 16), dict(recfield=1, dv=1, nfreq=10, addadj=True), lambda: models.counting_gnnml3(1, 12),
         models.counting_loss, 1, 'counting.py:335-372 (GNNML3, S = 12, 5 layers 16+16)'),
        ('sr25', 'regular', 64, (32, 32), dict(recfield=1, dv=1, nfreq=5), lambda: models.sr25_gnnml3(1, 6), None, 1,
         'sr25.py:248-280 (GNNML3, S = 6, 3 layers 32+16; synthetic 12-regular 25-node graphs)'),
        ('mutag_gnnml3', 'zinc', 512, (32, 16), dict(recfield=1, dv=1, nfreq=3), lambda: models.mutag_gnnml3(21, 4), models.mutag_loss, 21,
         'mutag.py:272-288 (GNNML3, S = 4, 3 layers 24+24, learnedge=False)'),
    ]


def _roofline(summ, nsteps):
    """dominant kernel among the launches that carry algorithmic bytes"""
    best = None
    for tag, v in summ.items():
        if v['bytes'] <= 0:
            continue
        per_step = v['ms'] * v['launches'] / nsteps
        if best is None or per_step > best[1]:
            best = (tag, per_step, v)
    if best is None:
        return None
    tag, per_step, v = best
    gbs = v['bytes'] / (v['ms'] * 1e-3) / 1e9
    return dict(bound='hbm', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s', frac=gbs / HBM_PEAK_GBS, traffic=None,
                kernel=tag, launches_per_step=v['launches'] / nsteps, avg_launch_ms=v['ms'], ms_per_step=per_step,
                algorithmic_bytes_per_launch=v['bytes'])


def run(dev, quick=False, only=None):
    from gnn_matlang_amd import SpectralDesign, collate, synthetic, functional as Fn
    out = []
    old_verbose = Fn.VERBOSE
    Fn.VERBOSE = True            # record the kernel family of every layer call (functional.PATHS)
    try:
        for name, kind, pool_n, reps, kw, ctor, loss, fdim, ref in _configs():
            if only and name not in only:
                continue
            raw = synthetic.make_graphs(kind, pool_n, seed=2)
            pool = SpectralDesign(**kw).design_many(raw)
            host = collate(pool * reps[1 if quick else 0])
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
            out.append(dict(config=name, reference=ref, graphs=B, nodes=int(data.x.size(0)), support_edges=int(data.edge_index2.size(1)),
                            S=int(data.edge_attr2.size(1)), ms_per_step=round(dt * 1e3, 3), graphs_per_s=round(B / dt),
                            roofline=_roofline(summ, n),
                            kernels_ms_per_step={k: round(v['ms'] * v['launches'] / n, 3) for k, v in summ.items()},
                            kernel_GBps={k: round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) for k, v in summ.items() if v['bytes'] > 0},
                            kernel_paths_per_step=paths,
                            slow_paths=sorted(k for k in paths if 'UNFUSED' in k or 'library GEMMs' in k or 'VALU kernels' in k)))
            del data, m, opt
    finally:
        Fn.VERBOSE = old_verbose
        Fn.PROFILE = None
    return out


if __name__ == '__main__':
    for rec in run(torch.device('cuda:0'), quick='--quick' in sys.argv):
        print(json.dumps(rec))
