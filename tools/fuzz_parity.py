#!/usr/bin/env python3
"""Randomised parity sweep of ML3Layer / SpectConv (HIP path) against the CPU oracle.

Not part of the pytest suites (it takes minutes): it draws random graph batches — isolated rows,
hubs above the staging caps, row counts that are not a multiple of the 128-row groups, duplicate
edges — and random layer shapes over every kernel family (8-wave forward, fused backward, the
generic kernels, matrix-core and VALU edge branch), and checks output and every gradient against
`oracle/spect_conv_oracle.py` evaluated in fp64.

    python tools/fuzz_parity.py [--cases 80] [--seed 0]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
TOL = 1e-4


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) if a.size else 0.0


def random_graph(rng, N, kind):
    if kind == 'molecule':
        deg = rng.integers(0, 5, N)
    elif kind == 'hubs':
        deg = rng.integers(0, 4, N)
        deg[rng.integers(0, N, max(1, N // 100))] = rng.integers(200, 1500)
    elif kind == 'dense':
        deg = rng.integers(20, 60, N)
    else:  # sparse with many isolated rows
        deg = (rng.random(N) < 0.3) * rng.integers(1, 3, N)
    dst = np.repeat(np.arange(N), deg)
    src = rng.integers(0, N, dst.size)
    if kind == 'molecule':  # keep sources near the row, as batched small graphs do
        src = np.clip(dst + rng.integers(-12, 13, dst.size), 0, N - 1)
    order = np.lexsort((dst, src))      # row-major COO as the reference's transform emits it
    return np.stack([src[order], dst[order]]).astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=80)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    from gnn_matlang_amd import ML3Layer
    from oracle.spect_conv_oracle import OracleML3Layer
    from oracle.relu_margin import make_safe
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(a.seed)
    from gnn_matlang_amd._lib import GmlError
    worst, fails, unsupported = 0.0, [], []
    for k in range(a.cases):
        kind = ['molecule', 'hubs', 'dense', 'sparse'][k % 4]
        N = int(rng.choice([1, 7, 63, 64, 127, 128, 129, 300, 777, 2048, 5000]))
        S = int(rng.choice([1, 2, 3, 4, 5, 8, 12]))
        learn = bool(rng.integers(0, 2))
        So = int(rng.choice([S, S, 4, 8, 3])) if learn else S
        Fin = int(rng.choice([1, 3, 8, 21, 32, 64, 70]))
        n1 = int(rng.choice([4, 16, 30, 32, 64]))
        n2 = int(rng.choice([0, 2, 10, 32]))
        ei = random_graph(rng, N, kind)
        E = ei.shape[1]
        torch.manual_seed(a.seed * 1000 + k)
        ref = OracleML3Layer(learn, S, So, Fin, n1, n2).double()
        m = ML3Layer(learn, S, So, Fin, n1, n2).to(dev)
        m.load_state_dict({n: p.detach().float() for n, p in ref.state_dict().items()})
        x = torch.randn(N, Fin)
        ea = torch.randn(E, S) * 0.5
        gout = torch.randn(N, n1 + n2)
        # keep every relu argument away from zero (oracle/relu_margin.py): a flipped mask is not a parity error
        ea, mask = make_safe(x, torch.from_numpy(ei), ea, ref.state_dict(), learn)
        gout[:, :n1] *= mask.float()
        xr, er = x.double().requires_grad_(True), ea.double().requires_grad_(True)
        yr = ref(xr, torch.from_numpy(ei), er)
        (yr * gout.double()).sum().backward()
        xg, eg = x.to(dev).requires_grad_(True), ea.to(dev).requires_grad_(True)
        tag = dict(case=k, kind=kind, N=N, E=int(E), S=S, So=So, learn=learn, Fin=Fin, n1=n1, n2=n2)
        try:
            y = m(xg, torch.from_numpy(ei).to(dev), eg)
            (y * gout.to(dev)).sum().backward()
        except (GmlError, NotImplementedError) as ex:      # a shape the library refuses loudly: listed, not a failure
            unsupported.append(tag)
            print('UNSUPPORTED', json.dumps(tag), str(ex)[:80], flush=True)
            continue
        torch.cuda.synchronize()
        errs = {'out': rel_err(y.detach().cpu(), yr.detach()), 'g_x': rel_err(xg.grad.cpu(), xr.grad)}
        if E:
            errs['g_ea'] = rel_err(eg.grad.cpu(), er.grad)
        gp = dict(m.named_parameters())
        for n, p in ref.named_parameters():
            errs[n] = rel_err(gp[n].grad.cpu(), p.grad)
        e = max(errs.values())
        worst = max(worst, e)
        if np.isfinite(e) and 0.3 * TOL < e <= TOL:
            print('NOTE', json.dumps(tag), {n: '%.1e' % v for n, v in errs.items() if v > 0.3 * TOL}, flush=True)
        if not np.isfinite(e) or e > TOL:
            fails.append((tag, {n: v for n, v in errs.items() if not v <= TOL}))
            print('FAIL', json.dumps(tag), fails[-1][1], flush=True)
    print(json.dumps({'cases': a.cases, 'seed': a.seed, 'worst_rel_err': worst, 'failures': len(fails), 'unsupported': len(unsupported),
                      'tol': TOL}))
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()
