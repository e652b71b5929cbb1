"""Train-step time of the other BASELINE configs (parity-test cases, not bench lines) on large synthetic batches, with the
per-tag kernel breakdown of functional.PROFILE, the kernel family every layer call took, and -- per config -- the same
`roofline` record bench.py carries for the ZINC step: dominant kernel, SURVEY s8(d) algorithmic bytes per launch, live mean
launch time (HIP events), fraction of the HBM roof.

    python tools/bench_configs.py [--quick] [--only=mutag_gnnml3]          one JSON line per config

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


def _mutag_like(count, seed):
    """ZINC-like molecular graphs with mutag's node features: one-hot over 7 atom types (libs/utils.py:192-209 reads 7 columns
    from mutag.mat); SpectralDesign(adddegree) appends the degree -> 8 input features like mutag.py:14,272-288"""
    import numpy as np
    from gnn_matlang_amd import synthetic
    rng = np.random.default_rng(seed)
    out = []
    for x, ei, y in synthetic.make_graphs('zinc', count, seed=seed):
        x7 = np.zeros((x.shape[0], 7), dtype=np.float32)
        x7[np.arange(x.shape[0]), rng.integers(7, size=x.shape[0])] = 1
        out.append((x7, ei, np.float32(rng.integers(2))))
    return out


def _sr25_real():
    from gnn_matlang_amd import readers
    return readers.load_sr(os.path.join(ROOT, 'tests', 'golden', 'raw', 'sr251256.g6'))


def _configs():
    from gnn_matlang_amd import models, synthetic
    # (name, graph pool, nodes wanted (full, quick), SpectralDesign kwargs of the reference script, model, loss, reference)
    return [
        ('counting', lambda: synthetic.make_graphs('counting', 512, seed=2), (600000, 150000),
         dict(recfield=1, dv=1, nfreq=10, adddegree=True, laplacien=False, addadj=True), lambda: models.counting_gnnml3(2, 12),
         models.counting_loss, 'counting.py:16,335-372 (GNNML3, S = 12, input [1, degree], 5 layers 16+16)'),
        ('sr25', _sr25_real, (600000, 150000), dict(recfield=1, dv=2, nfreq=5, adddegree=True), lambda: models.sr25_gnnml3(2, 6), None,
         'sr25.py:16,248-280 (GNNML3, S = 6, input [1, degree], 3 layers 32+16; the 15 REAL sr25 graphs, tiled)'),
        ('mutag_gnnml3', lambda: _mutag_like(512, 2), (600000, 150000), dict(recfield=1, dv=4, nfreq=3, adddegree=True), lambda: models.mutag_gnnml3(8, 4),
         models.mutag_loss, 'mutag.py:14,272-288 (GNNML3, S = 4, 7 atom types + degree, 3 layers 24+24 + BatchNorm, learnedge=False; ZINC-like synthetic graphs)'),
        ('mutag_gnnml1', lambda: _mutag_like(512, 2), (600000, 150000), dict(recfield=1, dv=4, nfreq=3, adddegree=True), lambda: models.GNNML1Mutag(8),
         models.mutag_loss, 'mutag.py:14,214-266 (GNNML1: three blocks [relu fc | relu SpectConv(K=1) | relu fc . relu fc] 16 + 32 + 16 + BatchNorm; BASELINE config 0\'s model; '
         'each block one fused launch, csrc/gml_gnnml1.hip)'),
        ('sr25_gnnml1', _sr25_real, (600000, 150000), dict(recfield=1, dv=2, nfreq=5, adddegree=True), lambda: models.sr25_gnnml1(2), None,
         'sr25.py:16,192-246 (GNNML1, sum form tanh(fc + SpectConv(K=1) + fc . fc), 64 wide, the 15 REAL sr25 graphs, tiled)'),
    ]


def _tile(base, reps, dev):
    """the collated pool `base` (on the device) repeated reps times as one block-diagonal batch"""
    from gnn_matlang_amd.graph import Batch
    n, B = base.x.size(0), base.num_graphs
    offs = torch.arange(reps, device=dev) * n
    return Batch(x=base.x.repeat(reps, 1),
                 edge_index=(base.edge_index.unsqueeze(1) + offs.view(1, -1, 1)).reshape(2, -1),
                 edge_index2=(base.edge_index2.unsqueeze(1) + offs.view(1, -1, 1)).reshape(2, -1),
                 edge_attr2=base.edge_attr2.repeat(reps, 1),
                 batch=(base.batch.unsqueeze(0) + (torch.arange(reps, device=dev) * B).view(-1, 1)).reshape(-1),
                 ptr=torch.cat([(base.ptr[:-1].long().view(1, -1) + offs.view(-1, 1)).reshape(-1),
                                torch.tensor([n * reps], device=dev)]).int(),
                 y=torch.zeros(B * reps, device=dev))


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
    from gnn_matlang_amd import SpectralDesign, collate, functional as Fn
    out = []
    old_verbose = Fn.VERBOSE
    Fn.VERBOSE = True            # record the kernel family of every layer call (functional.PATHS)
    try:
        for name, make_pool, nodes, kw, ctor, loss, ref in _configs():
            if only and name not in only:
                continue
            pool = SpectralDesign(**kw).design_many(make_pool())
            base = collate(pool).to(dev)
            want = nodes[1 if quick else 0]
            data = _tile(base, max((want + base.x.size(0) - 1) // base.x.size(0), 1), dev)
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
            n = 5
            t0 = time.perf_counter()                 # the step time: NO per-kernel events in the timed region (round 3 timed the
            for _ in range(n):                       # profiled pass below: its event pairs serialise the stream against the host --
                step()                               # 18.4 ms instead of 5.2 ms for the mutag config, whose torch BatchNorm kernels
            torch.cuda.synchronize()                 # otherwise overlap the host's launch work)
            dt = (time.perf_counter() - t0) / n
            Fn.PROFILE = {}
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            summ = Fn.profile_summary(Fn.PROFILE)
            Fn.PROFILE = None
            out.append(dict(config=name, reference=ref, graphs=B, nodes=int(data.x.size(0)), support_edges=int(data.edge_index2.size(1)),
                            S=int(data.edge_attr2.size(1)), ms_per_step=round(dt * 1e3, 3), graphs_per_s=round(B / dt),
                            roofline=_roofline(summ, n),
                            roofline_step=sum(v['bytes'] * v['launches'] for v in summ.values()) / n / dt / 1e9 / HBM_PEAK_GBS,
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
    only = [a.split('=', 1)[1] for a in sys.argv if a.startswith('--only=')]
    for rec in run(torch.device('cuda:0'), quick='--quick' in sys.argv, only=only or None):
        print(json.dumps(rec))
