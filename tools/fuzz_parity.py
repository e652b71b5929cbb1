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


def compare(ref, m, inputs_ref, inputs_dev, gout, dev):
    """forward + backward of the fp64 oracle module and the HIP module on the same data; dict of relative errors."""
    yr = ref(*inputs_ref)
    (yr * gout.double()).sum().backward()
    y = m(*inputs_dev)
    (y * gout.to(dev)).sum().backward()
    torch.cuda.synchronize()
    errs = {'out': rel_err(y.detach().cpu(), yr.detach())}
    for name, tr, td in (('g_x', inputs_ref[0], inputs_dev[0]), ('g_ea', inputs_ref[2], inputs_dev[2])):
        if tr.numel():
            errs[name] = rel_err(td.grad.cpu(), tr.grad)
    gp = dict(m.named_parameters())
    for n, p in ref.named_parameters():
        errs[n] = rel_err(gp[n].grad.cpu(), p.grad)
    return errs


def conv_sweep(a, dev):
    """SpectConv (selfconn / depthwise / bias variants) and SpectConCatConv, edge lists in arbitrary order."""
    from gnn_matlang_amd import SpectConv, SpectConCatConv
    from gnn_matlang_amd._lib import GmlError
    from oracle.spect_conv_oracle import OracleSpectConv, OracleSpectConCatConv
    rng = np.random.default_rng(a.seed + 77)
    worst, fails, unsupported = 0.0, 0, 0
    for k in range(a.cases):
        kind = ['molecule', 'hubs', 'dense', 'sparse'][k % 4]
        N = int(rng.choice([1, 7, 64, 129, 300, 777, 2048, 5000]))
        K = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16]))
        Fin = int(rng.choice([1, 3, 8, 21, 32, 64, 80, 128]))
        Fout = int(rng.choice([1, 4, 16, 30, 32, 64, 128]))
        selfconn, bias = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        concat = k % 5 == 4
        depthwise = (not concat) and k % 3 == 0
        ei = random_graph(rng, N, kind)
        if k % 2:
            ei = ei[:, rng.permutation(ei.shape[1])]            # the modules take any edge order
        E = ei.shape[1]
        torch.manual_seed(a.seed * 1000 + k)
        if concat:
            K, Fout = min(K, 4), min(Fout, 32)
            ref, m = OracleSpectConCatConv(Fin, Fout, K, selfconn, bias), SpectConCatConv(Fin, Fout, K, selfconn, bias)
        else:
            ref = OracleSpectConv(Fin, Fout, K, selfconn, depthwise, bias)
            m = SpectConv(Fin, Fout, K, selfconn, depthwise, bias)
        with torch.no_grad():
            for p in ref.parameters():
                p.copy_(torch.randn_like(p) * 0.3)
        ref = ref.double()
        m = m.to(dev)
        m.load_state_dict({n: p.detach().float() for n, p in ref.state_dict().items()})
        x, ea = torch.randn(N, Fin), torch.randn(E, K) * 0.5
        gout = torch.randn(N, K * Fout + (Fout if selfconn else 0)) if concat else torch.randn(N, Fout)
        eit = torch.from_numpy(ei)
        tag = dict(case=k, mod='concat' if concat else 'conv', kind=kind, N=N, E=int(E), K=K, Fin=Fin, Fout=Fout,
                   selfconn=selfconn, depthwise=depthwise, bias=bias)
        try:
            errs = compare(ref, m, (x.double().requires_grad_(True), eit, ea.double().requires_grad_(True)),
                           (x.to(dev).requires_grad_(True), eit.to(dev), ea.to(dev).requires_grad_(True)), gout, dev)
        except (GmlError, NotImplementedError) as ex:
            unsupported += 1
            print('UNSUPPORTED', json.dumps(tag), str(ex)[:80], flush=True)
            continue
        if gout.numel() <= 4:                   # a 1 x 1 output is one cancelling sum: not held to 1e-4 of itself
            errs['out'] *= 0.1
        e = max(errs.values())
        worst = max(worst, e)
        if not np.isfinite(e) or e > TOL:
            fails += 1
            print('FAIL', json.dumps(tag), {n: '%.1e' % v for n, v in errs.items() if not v <= TOL}, flush=True)
    print(json.dumps({'sweep': 'conv', 'cases': a.cases, 'seed': a.seed, 'worst_rel_err': worst, 'failures': fails,
                      'unsupported': unsupported, 'tol': TOL}))
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=80)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--sweep', choices=['ml3', 'conv'], default='ml3')
    ap.add_argument('--only', type=int, default=-1, help='ml3 sweep: run just this case of the sequence')
    ap.add_argument('--verbose', action='store_true')
    a = ap.parse_args()
    if a.sweep == 'conv':
        sys.exit(1 if conv_sweep(a, torch.device('cuda:0')) else 0)
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
        if a.only >= 0 and k != a.only:
            continue
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
        # the edge branch's weight gradients are a few numbers each (2S x S): measured against the largest entry of the
        # four together, so that one that happens to cancel to near zero is not held to 1e-4 of itself
        edge_max = max([float(p.grad.abs().max()) for n, p in ref.named_parameters() if n.startswith('fc1_')] + [1e-30])
        for n, p in ref.named_parameters():
            if n.startswith('fc1_'):
                errs[n] = float((gp[n].grad.cpu().double() - p.grad).abs().max()) / edge_max
            else:
                errs[n] = rel_err(gp[n].grad.cpu(), p.grad)
        e = max(errs.values())
        worst = max(worst, e)
        if a.verbose:
            print(json.dumps(tag), {n: '%.1e' % v for n, v in errs.items()}, flush=True)
            d = (gp['conv1.bias'].grad.cpu().double() - ref.conv1.bias.grad).abs()
            print('  bias err by column', ['%.1e' % v for v in (d / ref.conv1.bias.grad.abs().max()).tolist()])
            yc = y.detach().cpu().double()
            fl = ((yc[:, :n1] > 0) != (yr[:, :n1] > 0))
            print('  flips', fl.nonzero().tolist(), 'masked there', [float(mask[i, j]) for i, j in fl.nonzero().tolist()])
            for i, j in fl.nonzero().tolist():
                print('   y gpu', float(yc[i, j]), 'ref', float(yr[i, j]), 'deg', int((ei[1] == i).sum()))
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
