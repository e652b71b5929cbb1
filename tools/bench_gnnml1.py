"""Train-step time of the GNNML1 models (sr25.py:192-246 sum form, mutag.py:214-266 factor form) with the block as one fused launch
(csrc/gml_gnnml1.hip) and, with GML_NO_GNNML1_FUSED=1, as rounds 1-4 ran it (library Linears + one S = 1 SpectConv + elementwise ops).

    python tools/bench_gnnml1.py [--graphs 24000]            one JSON line per model"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_matlang_amd import collate, models, synthetic            # noqa: E402


def batch(kind, graphs, dev):
    raw = synthetic.make_graphs('zinc', min(graphs, 2048), seed=7)
    rng = np.random.default_rng(0)
    items = []
    for x, ei, y in raw:
        n = x.shape[0]
        if kind == 'mutag':
            f = np.zeros((n, 8), dtype=np.float32)
            f[np.arange(n), rng.integers(7, size=n)] = 1
            f[:, 7] = np.bincount(ei[0], minlength=n)
        else:
            f = np.stack([np.ones(n, dtype=np.float32), np.bincount(ei[0], minlength=n).astype(np.float32)], 1)
        items.append(dict(x=f, edge_index=ei, y=np.float32(rng.integers(2))))
    reps = (graphs + len(items) - 1) // len(items)
    return collate(items * reps).to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--graphs', type=int, default=24000)
    ap.add_argument('--steps', type=int, default=20)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    for name, kind, make in (('sr25_gnnml1 (sum form, tanh, 64 wide)', 'sr25', lambda: models.sr25_gnnml1(2)),
                             ('mutag_gnnml1 (factor form, relu, 16 + 32 + 16, BatchNorm)', 'mutag', lambda: models.GNNML1Mutag(8))):
        b = batch(kind, a.graphs, dev)
        torch.manual_seed(0)
        m = make().to(dev)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        tgt = torch.randn(b.num_graphs, device=dev)

        def step():
            opt.zero_grad(set_to_none=True)
            out = m(b)
            loss = (out[:, 0] - tgt).abs().sum()
            loss.backward()
            opt.step()
            return loss
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        print(json.dumps(dict(model=name, fused=not os.environ.get('GML_NO_GNNML1_FUSED'), graphs=int(b.num_graphs), nodes=int(b.x.size(0)),
                              edges=int(b.edge_index.size(1)), ms_per_step=round(ms, 3), graphs_per_s=round(b.num_graphs / ms * 1e3))))


if __name__ == '__main__':
    main()
