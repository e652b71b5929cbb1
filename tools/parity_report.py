#!/usr/bin/env python3
"""Per-tensor parity errors (max|got - ref| / max|ref|) of the model fixtures in BOTH projection arithmetics
(default bf16x3 split, exact f32-input MFMA): the numbers behind the tolerances of tests/test_gpu_parity.py.
    python tools/parity_report.py > profiles/r02_parity_report.jsonl"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import Golden, rel_err            # noqa: E402
from gnn_matlang_amd import functional as Fn, models   # noqa: E402
from gnn_matlang_amd.graph import Batch          # noqa: E402

T = lambda a: torch.tensor(np.asarray(a))
dev = torch.device('cuda:0')


def batch_from(g):
    b = g.sub('batch/')
    ptr = np.concatenate([[0], np.cumsum(np.bincount(b['batch']))]).astype(np.int32)
    return Batch(x=T(b['x']), edge_index=T(b['edge_index']), edge_index2=T(b['edge_index2']),
                 edge_attr2=T(b['edge_attr2']), batch=T(b['batch']), ptr=T(ptr), y=T(b['y'])).to(dev)


for mode in ('bf16x3', 'f32'):
    Fn.F32_MFMA = mode == 'f32'
    for fname, ctor, loss in (('model_zinc_gnnml3.npz', 'zinc_gnnml3', 'zinc_loss'),
                              ('model_counting_gnnml3.npz', 'counting_gnnml3', 'counting_loss'),
                              ('model_mutag_gnnml3.npz', 'mutag_gnnml3', 'mutag_loss'),
                              ('model_mutag_gnnml1.npz', 'GNNML1Mutag', 'mutag_loss')):
        g = Golden(fname)
        data = batch_from(g)
        m = getattr(models, ctor)(8) if ctor == 'GNNML1Mutag' else getattr(models, ctor)()
        m.load_state_dict({k: T(v) for k, v in g.sub('param/').items()})
        m = m.to(dev).train()
        lf = getattr(models, loss)
        pre = m(data)
        l = lf(pre, data.y)
        l.backward()
        errs = {'logits': rel_err(pre.detach().cpu(), g['logits']), 'loss': abs(l.item() - float(g['loss'])) / abs(float(g['loss']))}
        for n, p in m.named_parameters():
            errs['grad ' + n] = rel_err(p.grad.cpu(), g['grad/' + n])
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        traj = []
        for _ in range(5):
            opt.zero_grad()
            l = lf(m(data), data.y)
            l.backward()
            opt.step()
            traj.append(l.item())
        errs['loss_traj'] = float(np.max(np.abs(np.array(traj) - g['loss_traj']) / np.abs(g['loss_traj'])))
        worst = max(errs, key=errs.get)
        print(json.dumps(dict(fixture=fname, mode=mode, worst=worst, worst_err=errs[worst],
                              over_1e4={k: float('%.3g' % v) for k, v in errs.items() if v > 1e-4},
                              over_5e5=sorted(k for k, v in errs.items() if v > 5e-5))), flush=True)
Fn.F32_MFMA = False
