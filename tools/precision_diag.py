#!/usr/bin/env python3
"""Where does the headline arithmetic's trained-state gradient error come from?  (VERDICT r05 "what's weak" #1)

bench.py's batch (131,072 ZINC-like graphs = 2,048 x 64), the ZINC GNNML3 trained for --steps Adam steps in the default
arithmetic, then ONE forward + backward per configuration of (forward arithmetic, backward arithmetic, kernel families taken exact)
against the float64 oracle under the term-sum criterion (oracle/parity_at_size.py) -- the same check as
tests/test_gpu_parity.py::test_bench_size_train_step_vs_fp64_oracle.  One JSON line per configuration on stdout."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, nargs='+', default=[100, 400])
    ap.add_argument('--batch', type=int, default=131072)
    ap.add_argument('--pool', type=int, default=2048)
    ap.add_argument('--seed', type=int, default=1000)
    ap.add_argument('--configs', nargs='*', default=None)
    a = ap.parse_args()
    import bench
    from gnn_matlang_amd import functional as Fn, models
    from oracle import parity_at_size as PS
    dev = torch.device('cuda:0')
    full, base = bench.build_batch(a.batch, a.pool, a.seed, dev)
    host = base.to(torch.device('cpu'))
    torch.manual_seed(0)
    m = models.zinc_gnnml3().to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    configs = {           # name: (forward exact, backward exact, families or None)
        'default': (False, False, None),
        'exact': (True, True, None),
        'fwd_exact': (True, False, None),
        'bwd_exact': (False, True, None),
        'edge_exact': (True, True, {'edge'}),
        'conv_exact': (True, True, {'conv'}),
        'fwd_edge_exact': (True, False, {'edge'}),
        'fwd_conv_exact': (True, False, {'conv'}),
    }
    if os.environ.get('GML_DIAG_MODES'):      # extra launch-flag modes under test, e.g. "f16x3"
        for name in os.environ['GML_DIAG_MODES'].split(','):
            configs[name] = name
    names = a.configs or list(configs)
    done = 0
    for target in sorted(a.steps):
        while done < target:
            opt.zero_grad(set_to_none=True)
            models.zinc_loss(m(full), full.y).backward()
            opt.step()
            done += 1
        T = None
        for name in names:
            cfg = configs[name]
            m.zero_grad()
            if isinstance(cfg, str):
                ctxm = Fn.product_mode(cfg)
                fe = be = False
            else:
                fe, be, fam = cfg
                Fn.EXACT_ONLY, Fn.FORCE_BWD_EXACT = fam, be
                ctxm = Fn.exact_products(fe)
            try:
                with ctxm:
                    cap = {}
                    pre = m(full, _capture=cap)
                    loss = models.zinc_loss(pre, full.y)
                loss.backward()
            finally:
                Fn.EXACT_ONLY, Fn.FORCE_BWD_EXACT = None, None
            torch.cuda.synchronize()
            ref = PS.reference(host, m.state_dict(), full.y, pre_dev=pre[:, 0], T=T, head_pre_dev=cap['head_pre'])
            T = ref['T']
            rep = PS.compare(ref, pre[:, 0].detach().cpu().numpy(), {n: p.grad.detach().cpu().numpy() for n, p in m.named_parameters()}, tol=1e-4)
            worst = sorted(rep['tensors'].items(), key=lambda kv: -kv[1]['termsum'])[:3]
            print(json.dumps(dict(steps=done, config=name, logits=rep['logits_rel_err'], termsum=rep['worst_termsum'], maxnorm=rep['worst_maxnorm'],
                                  head_units_flipped=ref['head_units_flipped'],
                                  worst=[(n, float('%.3g' % v['termsum'])) for n, v in worst])), flush=True)


if __name__ == '__main__':
    main()
