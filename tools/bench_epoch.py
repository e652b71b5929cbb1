"""The captured batch-64 epoch of bench.py (`epoch_bs64`) alone, for a kernel trace of ONE replayed step:

    rocprofv3 --kernel-trace --stats -d gpurun_out/epoch -- python3 tools/bench_epoch.py [--torch-assembly] [--replays 200]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--torch-assembly', action='store_true')
    ap.add_argument('--replays', type=int, default=200)
    ap.add_argument('--graphs', type=int, default=2000)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--plain-head', action='store_true', help='head + loss through library calls (rounds 1-4)')
    args = ap.parse_args()
    from gnn_matlang_amd import SpectralDesign, models, synthetic, functional as Fn
    from gnn_matlang_amd.dataset import DeviceDataset
    from gnn_matlang_amd.optim import OneLaunchAdam
    dev = torch.device('cuda:0')
    raw = synthetic.make_graphs('zinc', args.graphs, seed=4242)
    dsd = DeviceDataset.from_graphs(SpectralDesign(recfield=2, dv=2, nfreq=7).design_many(raw), dev)
    dsd.y = dsd.y.float()
    dsd.prepare()
    bd = dsd.bounds(args.batch)
    Bq, G_ = args.batch, len(dsd)
    ids_buf = torch.zeros(Bq, dtype=torch.int64, device=dev)
    assemble = dsd.batch_padded if args.torch_assembly else dsd.batch_assembled
    torch.manual_seed(0)
    cm = models.zinc_gnnml3().to(dev)
    co = OneLaunchAdam(cm.parameters(), lr=1e-3)
    loss_acc = torch.zeros((), device=dev)
    one_ = torch.ones((), device=dev)

    def padded_step():
        b = assemble(ids_buf, bd)
        co.zero_grad(set_to_none=True)
        if args.plain_head:
            pre = cm(b)
            l = ((pre[:Bq, 0] - b.y[:Bq]).abs() * b.graph_valid).sum()
        else:
            l = models.zinc_step_loss(cm, b, loss_sum=loss_acc)       # (the epoch's running loss: accumulated inside the head's launch)
        if args.plain_head:
            l.backward()
        else:
            with Fn.deferred_folds(None):
                l.backward(one_)
        co.step()
        if args.plain_head:
            loss_acc.add_(l.detach())
    ids_buf.copy_(torch.arange(Bq, device=dev))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            padded_step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        padded_step()
    gen = torch.Generator().manual_seed(7)
    perm = torch.randperm(G_, generator=gen).to(dev)
    for _ in range(20):
        cg.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(args.replays):
        i = (r * Bq) % (G_ - Bq)
        ids_buf.copy_(perm[i:i + Bq])
        cg.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.replays
    print(json.dumps(dict(ms_per_step=dt * 1e3, n_pad=bd['n_pad'], e2_pad=bd['e2_pad'], dmax=bd['dmax'], replays=args.replays,
                          assembly='torch' if args.torch_assembly else 'gml_batch_assemble')))


if __name__ == '__main__':
    main()
