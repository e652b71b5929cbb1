"""Where does the headline mode's error at bench size come from?  (test driver: imports the oracle)
Trains the bench's model for --steps Adam steps on the bench's batch, then prints, per arithmetic mode, every parameter gradient's
worst |err| / T (layer-local term sums) and every layer's activation error against the float64 oracle."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from gnn_matlang_amd import functional as Fn, models              # noqa: E402
from oracle import models_oracle as MO, parity_at_size as PS      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--graphs', type=int, default=131072)
    ap.add_argument('--pool', type=int, default=2048)
    ap.add_argument('--steps', type=int, default=400)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    full, base = bench.build_batch(a.graphs, a.pool, 1000, dev)
    torch.manual_seed(0)
    m = models.zinc_gnnml3().to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    for _ in range(a.steps):
        opt.zero_grad(set_to_none=True)
        models.zinc_loss(m(full), full.y).backward()
        opt.step()
    host = base.to(torch.device('cpu'))
    P = int(host.num_graphs)
    # float64 activations of the pool
    mo = MO.zinc_gnnml3(int(host.x.size(1)), int(host.edge_attr2.size(1))).double()
    mo.load_state_dict({k: v.detach().cpu().double() for k, v in m.state_dict().items()})
    acts, live = {}, {}
    hs = [getattr(mo, 'conv%d' % i).register_forward_hook(lambda mod, inp, out, i=i: (acts.__setitem__(i, out.detach()), live.__setitem__(i, out)) and None) for i in range(1, 5)]
    pre64 = mo(host.x.double(), host.edge_index2, host.edge_attr2.double(), host.batch, P)[:, 0]
    for h in hs:
        h.remove()
    R = a.graphs // P
    cg = torch.sign(pre64.detach().unsqueeze(0) - full.y.cpu().double().view(R, P)).sum(0)
    gouts = dict(zip(range(1, 5), torch.autograd.grad((cg * pre64).sum(), [live[i] for i in range(1, 5)])))
    nout1 = {i: int(getattr(mo, 'conv%d' % i).conv1.weight.size(2)) for i in range(1, 5)}
    n0 = int(host.x.size(0))
    T = None
    for mode in ('bf16x3', 'f32'):
        dacts = {}
        hs = [getattr(m, 'conv%d' % i).register_forward_hook(lambda mod, inp, out, i=i: dacts.__setitem__(i, out.detach())) for i in range(1, 5)]
        for p in m.parameters():
            p.grad = None
        with Fn.exact_products(mode == 'f32'):
            cap = {}
            pre = m(full, _capture=cap)
            models.zinc_loss(pre, full.y).backward()
        for h in hs:
            h.remove()
        # where the forward error enters: layer 1's edge branch alone (device, both kernel families) and layer 1's conv / Hadamard columns
        with torch.no_grad(), Fn.exact_products(mode == 'f32'):
            l1 = m.conv1
            e = base.edge_attr2.contiguous()
            W = [l1.fc1_1.weight, l1.fc1_2.weight, l1.fc1_3.weight, l1.fc1_4.weight]
            if mode == 'f32':
                ead, _ = Fn.edge_mlp_fwd(e, *W)
            else:
                ead, _ = Fn.edge_mlp_fwd(e, *W, None, Fn.edge_presplit(e))
            e64 = e.cpu().double()
            W64 = [w.detach().cpu().double() for w in W]
            h = torch.cat([torch.relu(e64 @ W64[0].t()), torch.tanh(e64 @ W64[1].t()) * torch.tanh(e64 @ W64[2].t())], 1)
            ea64 = torch.relu(h @ W64[3].t())
            e32 = e.cpu().float()
            W32 = [w.detach().cpu().float() for w in W]
            h32 = torch.cat([torch.relu(e32 @ W32[0].t()), torch.tanh(e32 @ W32[1].t()) * torch.tanh(e32 @ W32[2].t())], 1)
            ea32 = torch.relu(h32 @ W32[3].t())
            d = (ead.cpu().double() - ea64).abs()
            d32 = (ea32.double() - ea64).abs()
            print('   layer-1 edge branch: device max err %.2e rms %.2e | torch fp32 CPU max err %.2e rms %.2e  (max |ea| %.2e)' % (
                d.max(), d.pow(2).mean().sqrt(), d32.max(), d32.pow(2).mean().sqrt(), ea64.abs().max()))
        ref = PS.reference(host, m.state_dict(), full.y, pre_dev=pre[:, 0], T=T, head_pre_dev=cap['head_pre'])
        T = ref['T']
        rep = PS.compare(ref, pre[:, 0].detach().cpu().numpy(), {n: p.grad.detach().cpu().numpy() for n, p in m.named_parameters()})
        print('== %s  logits %.2e  worst termsum %.2e  maxnorm %.2e  head units flipped %d' % (mode, rep['logits_rel_err'], rep['worst_termsum'], rep['worst_maxnorm'], ref['head_units_flipped']))
        for i in range(1, 5):
            if i in dacts and dacts[i].size(0) >= 3 * n0:
                print('   layer %d: copies 0 / 1 / last of the device output bit-identical: %s %s' % (
                    i, bool(torch.equal(dacts[i][:n0], dacts[i][n0:2 * n0])), bool(torch.equal(dacts[i][:n0], dacts[i][-n0:]))))
            if i in dacts and dacts[i].size(0) >= n0:
                d = dacts[i][:n0].cpu().double()
                e = (d[:, :acts[i].size(1)] - acts[i]).abs()
                c1 = nout1[i]
                mism = (d[:, :c1] > 0) != (acts[i][:, :c1] > 0)
                ga = gouts[i][:, :c1].abs()
                small = [(int(((acts[i][:, :c1] > 0) & (acts[i][:, :c1] < t)).sum())) for t in (1e-7, 1e-6, 1e-5, 1e-4)]
                print('   layer %d relu units: %d of %d differ from the float64 mask (copy 0); sum |g| there / sum |g| over live units = %.2e worst column %.2e;  live units below 1e-7/1e-6/1e-5/1e-4: %s' % (
                    i, int(mism.sum()), mism.numel(), float((ga * mism).sum() / (ga * (acts[i][:, :c1] > 0)).sum()),
                    float(((ga * mism).sum(0) / (ga * (acts[i][:, :c1] > 0)).sum(0).clamp(min=1e-300)).max()), small))
                ec, eh = e[:, :c1], e[:, c1:]
                print('   act %d columns: conv max err %.2e rms %.2e | Hadamard max err %.2e rms %.2e' % (i, ec.max(), ec.pow(2).mean().sqrt(),
                      eh.max() if eh.numel() else 0.0, eh.pow(2).mean().sqrt() if eh.numel() else 0.0))
                print('   act %d: max err %.2e of max |x| %.2e  (rel %.2e), rms err / rms x %.2e' % (
                    i, e.max(), acts[i].abs().max(), e.max() / acts[i].abs().max(), e.pow(2).mean().sqrt() / acts[i].pow(2).mean().sqrt()))
        if mode == 'f32':
            fp32_oracle_line(host, m, full, pre[:, 0], T)
        for n, v in sorted(rep['tensors'].items(), key=lambda kv: -kv[1]['termsum']):
            g = ref['grads'][n]
            print('   %-22s termsum %.2e  maxnorm %.2e   median T/|g| %.0f' % (n, v['termsum'], v['maxnorm'],
                  np.median(ref['T'][n] / np.maximum(np.abs(g), 1e-300))))


def fp32_oracle_line(host, m, full, pre_dev, T):
    """the reference arithmetic (oracle, float32, CPU) under the same criterion"""
    pre32, g32, z32 = PS.oracle_fp32_as_device(host, m.state_dict(), full.y, pre_dev)
    ref = PS.reference(host, m.state_dict(), full.y, pre_dev=pre_dev, T=T, head_pre_dev=z32)
    rep = PS.compare(ref, pre32, g32)
    worst = max(rep['tensors'].items(), key=lambda kv: kv[1]['termsum'])
    print('== oracle fp32 (CPU)  logits %.2e  worst termsum %.2e (%s)  maxnorm %.2e  head units flipped %d' % (
        rep['logits_rel_err'], rep['worst_termsum'], worst[0], rep['worst_maxnorm'], ref['head_units_flipped']))


if __name__ == '__main__':
    main()
