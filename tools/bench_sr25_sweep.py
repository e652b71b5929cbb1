"""Config 5 as SURVEY s8(d) "R" specifies it: the 15 REAL sr25 graphs (tests/golden/raw/sr251256.g6 through readers.load_sr),
tiled x T to >= 1 M nodes, S in {6, 12, 24, 48} supports (SpectralDesign(recfield=1, dv=2, nfreq=S-1, adddegree) -- sr25.py:16
with nfreq raised), forward only like sr25.py:282-300, plus the stand-alone multi-support SpMM (BASELINE.json's "SpMM HBM GB/s
vs peak"), GB/s against the 8 TB/s roof per S.

    python tools/bench_sr25_sweep.py [--nodes 1000000] [--S 6,12,24,48]        one JSON line per S

bench.py embeds `run(dev, quick=True)` as its `sr25_sweep` block."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def _median_launch_ms(fn, reps=10, blocks=5):
    for _ in range(2):
        fn()
    t = []
    for _ in range(blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) / reps)
    return float(np.median(t))


def run(dev, nodes=1000000, supports=(6, 12, 24, 48), quick=False):
    from gnn_matlang_amd import SpectralDesign, collate, models, readers, functional as Fn
    from gnn_matlang_amd.graph import Batch
    raw = readers.load_sr(os.path.join(ROOT, 'tests', 'golden', 'raw', 'sr251256.g6'))
    if quick:
        nodes = min(nodes, 250000)
    out = []
    for S in supports:
        ds = SpectralDesign(recfield=1, dv=2, nfreq=S - 1, adddegree=True).design_many(raw)     # sr25.py:16 (nfreq = 5 there)
        base = collate(ds).to(dev)
        n, B = base.x.size(0), base.num_graphs
        reps = (nodes + n - 1) // n
        offs = torch.arange(reps, device=dev) * n
        data = Batch(x=base.x.repeat(reps, 1),
                     edge_index=(base.edge_index.unsqueeze(1) + offs.view(1, -1, 1)).reshape(2, -1),
                     edge_index2=(base.edge_index2.unsqueeze(1) + offs.view(1, -1, 1)).reshape(2, -1),
                     edge_attr2=base.edge_attr2.repeat(reps, 1),
                     batch=(base.batch.unsqueeze(0) + (torch.arange(reps, device=dev) * B).view(-1, 1)).reshape(-1),
                     ptr=torch.cat([(base.ptr[:-1].long().view(1, -1) + offs.view(-1, 1)).reshape(-1),
                                    torch.tensor([n * reps], device=dev)]).int(),
                     y=torch.zeros(B * reps, device=dev))
        csr = data.csr('edge_index2')
        N, E = csr.N, csr.E
        rec = dict(S=S, graphs=data.num_graphs, nodes=N, support_edges=E, nnz_per_row=E / N)
        # ---- stand-alone SpMM H = [A_s^T X]_s at the layer widths of sr25.py:252-262 (Fin = 48 hidden; 32 = the 8-wave kernel's width)
        vals = csr.sort_values(data.edge_attr2)
        for Fin in (32, 48):
            xs = torch.randn(N, Fin, device=dev)
            ms = _median_launch_ms(lambda: Fn.spmm(csr, vals, xs, S, Fin))
            q = 4 * (E * S + N * Fin + N * S * Fin) + 4 * (E + N + 1)
            rec['spmm_Fin%d' % Fin] = dict(avg_launch_ms=ms, algorithmic_bytes_per_launch=q, achieved=q / (ms * 1e-3) / 1e9,
                                           peak=HBM_PEAK_GBS, unit='GB/s', frac=q / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, bound='hbm')
            del xs
        # ---- the model forward (sr25.py:282-300: eval mode, no gradients), fused layers
        torch.manual_seed(0)
        m = models.sr25_gnnml3(2, S).to(dev).eval()
        Fn.PROFILE = None
        with torch.no_grad():
            ms_f = _median_launch_ms(lambda: m(data), reps=3, blocks=3)
            Fn.PROFILE = {}
            for _ in range(3):
                m(data)
            torch.cuda.synchronize()
            summ = Fn.profile_summary(Fn.PROFILE)
            Fn.PROFILE = None
        if S <= 8 and Fn.FWD_CHUNKS:
            # A/B: the same forward with GML_FWD_CHUNKS=0 -- groups beyond one work item on the 64-row kernel family (rounds 1-3)
            Fn.FWD_CHUNKS = False
            try:
                with torch.no_grad():
                    ms_c = _median_launch_ms(lambda: m(data), reps=3, blocks=3)
                    Fn.PROFILE = {}
                    for _ in range(3):
                        m(data)
                    torch.cuda.synchronize()
                    sc = Fn.profile_summary(Fn.PROFILE)
                rec['forward_64row_family'] = dict(ms=ms_c, graphs_per_s=data.num_graphs / (ms_c * 1e-3), note='GML_FWD_CHUNKS=0 (not the default)',
                                                   kernels={k: dict(ms=round(v['ms'], 4), frac=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v['bytes'] else None)
                                                            for k, v in sc.items()})
            finally:
                Fn.FWD_CHUNKS = True
                Fn.PROFILE = None
        rec['forward'] = dict(ms=ms_f, graphs_per_s=data.num_graphs / (ms_f * 1e-3),
                              kernels={k: dict(ms=round(v['ms'], 4), launches_per_forward=v['launches'] / 3,
                                               GBps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['bytes'] else None,
                                               frac=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v['bytes'] else None)
                                       for k, v in summ.items()})
        out.append(rec)
        del data, csr, vals, m
        torch.cuda.empty_cache()
    return out


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--nodes', type=int, default=1000000)
    ap.add_argument('--S', default='6,12,24,48')
    a = ap.parse_args()
    for r in run(torch.device('cuda:0'), a.nodes, tuple(int(s) for s in a.S.split(','))):
        print(json.dumps(r))
