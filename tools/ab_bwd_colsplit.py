"""VERDICT r03 item 2, the cheap host-level A/B: the ZINC layer's fused backward (S = 8, Fin = 32, Fout = 30, learned supports:
dX + dval + dW) as ONE launch of gml_k_spectconv_bwd3<8, 2, 8, .., NOB = 2> against TWO launches of the NOB = 1 instantiation over
the output-column halves [0, 16) and [16, 30) (G / W column slices; dX accumulates with GML_ACCUM, dval with GML_DVAL_ACCUM, dW
splits by column).  If two column passes inside one group iteration are to pay, the NOB = 1 launch must cost well under half the
NOB = 2 launch's compute phases -- this measures it, with the results checked against each other.

    python tools/ab_bwd_colsplit.py [--graphs 131072]      ->  text for profiles/r04_bwd_colsplit_ab.txt
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--graphs', type=int, default=131072)
    args = ap.parse_args()
    import bench_configs as bc
    from gnn_matlang_amd import SpectralDesign, collate, synthetic, functional as Fn, _lib
    from gnn_matlang_amd.functional import _ptr, _off, _stream
    dev = torch.device('cuda:0')
    pool = SpectralDesign(recfield=2, dv=2, nfreq=7).design_many(synthetic.make_graphs('zinc', 2048, seed=3))
    base = collate(pool).to(dev)
    data = bc._tile(base, max(args.graphs // base.num_graphs, 1), dev)
    csr = data.csr('edge_index2')
    N, E, S, Fin, Fout = csr.N, csr.E, 8, 32, 30
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn(N, Fin, generator=g).to(dev)
    val_t = torch.rand(E, S, generator=g).to(dev)                       # source-order supports
    w = (torch.randn(S, Fin, Fout, generator=g) * 0.2).to(dev)
    G = torch.zeros(N, 32, device=dev)
    G[:, :Fout] = torch.randn(N, Fout, generator=g).to(dev)
    Gv = G[:, :Fout]

    def one_launch():
        return Fn.fused_conv_bwd(csr, val_t, x, Gv, w, True, True, True, None, None, 0)

    def plan(fo):
        p = Fn._bwd_plan(csr, S, Fin, fo)
        assert p is not None and p[4] == 128, p
        return p
    halves = ((0, 16), (16, Fout))
    plans = [plan(o1 - o0) for o0, o1 in halves]
    wp = [w[:, :, o0:o1].contiguous() for o0, o1 in halves]

    def two_launches():
        dx = torch.empty(N, Fin, device=dev)
        dval = torch.empty(E, S, device=dev)
        dws = []
        for part, (o0, o1) in enumerate(halves):
            flags, ginfo, gmax, nbytes, _ = plans[part]
            if part == 1:
                flags |= _lib.GML_ACCUM | _lib.GML_DVAL_ACCUM
            dw_p = torch.empty(S, Fin, o1 - o0, device=dev)
            ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
            _lib.call('gml_spectconv_bwd', _ptr(csr.rowptr_t), _ptr(csr.col_t), _ptr(ginfo), _ptr(val_t), _ptr(x), int(x.stride(0)),
                      _off(G, o0), int(G.stride(0)), _ptr(wp[part]), _ptr(dx), Fin, _ptr(dval), _ptr(dw_p), N, S, Fin, o1 - o0,
                      gmax[0], gmax[1], flags, _ptr(ws), ws.numel(), _stream(dev))
            dws.append(dw_p)
        return dx, dval, torch.cat(dws, 2)

    def time_ms(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    r1, r2 = one_launch(), two_launches()
    errs = [float((p - q).abs().max() / q.abs().max()) for p, q in zip(r2, r1)]
    t1, t2 = time_ms(one_launch), time_ms(two_launches)
    # one NOB = 1 launch alone (the first half), for the per-pass cost
    def first_half():
        flags, ginfo, gmax, nbytes, _ = plans[0]
        dx = torch.empty(N, Fin, device=dev); dval = torch.empty(E, S, device=dev); dw_p = torch.empty(S, Fin, 16, device=dev)
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
        _lib.call('gml_spectconv_bwd', _ptr(csr.rowptr_t), _ptr(csr.col_t), _ptr(ginfo), _ptr(val_t), _ptr(x), int(x.stride(0)),
                  _ptr(G), int(G.stride(0)), _ptr(wp[0]), _ptr(dx), Fin, _ptr(dval), _ptr(dw_p), N, S, Fin, 16,
                  gmax[0], gmax[1], flags, _ptr(ws), ws.numel(), _stream(dev))
    th = time_ms(first_half)
    q, _ = Fn.conv_cost_bwd(N, E, S, Fin, Fout, True, True)
    print('ZINC layer backward, %d graphs: N = %d rows, E = %d support edges, S = 8, Fin = 32, Fout = 30 (dX + dval + dW)' % (data.num_graphs, N, E))
    print('  one launch,  bwd3<8, 2, 8, NOB = 2> (2 waves / SIMD)              : %.3f ms   = %.3f of the HBM roof (%.0f MB algorithmic)' % (t1, q / (t1 * 1e-3) / 8e12, q / 1e6))
    print('  two launches, bwd3<8, 2, 8, NOB = 1> over columns [0,16) + [16,30) : %.3f ms   (%.2f x)' % (t2, t2 / t1))
    print('  one NOB = 1 launch alone (columns [0,16))                           : %.3f ms   (%.2f of the NOB = 2 launch)' % (th, th / t1))
    print('  two-launch results vs one launch (max |diff| / max |ref|): dX %.1e  dval %.1e  dW %.1e' % tuple(errs))


if __name__ == '__main__':
    main()
