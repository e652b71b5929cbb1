#!/usr/bin/env python3
"""bench.py -- GNNML3 training-step throughput on ZINC-12k-shaped synthetic graphs (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--batch GRAPHS_PER_GPU]

A step = forward + L1-sum loss + backward + (N>1: one flat SUM all-reduce of the gradients) + Adam,
of the reference's ZINC GNNML3 (Zinc12k.py:310-371: 4 x ML3Layer 30+2, S = 8 supports, 25 input
features, add-pool, fc 32 -> 1, lr 1e-3) over one batch of synthetic ZINC-like graphs per GPU (weak
scaling: graphs per GPU fixed).  Inputs are resident in HBM before the timed region.  fp32.
Rank 0 prints ONE JSON line; ``roofline`` is for the dominant kernel (the fused SpectConv backward; the
fused forward is under ``roofline_other``), timed live with HIP events inside the timed region; ``cpu_baseline`` is the CPU oracle (a port of the
reference algorithm, op for op) timed on this box's host cores on a bounded sample (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured achievable
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-in MFMA (= fp32 vector) peak


def build_batch(graphs_per_gpu, pool, seed, device):
    """pool distinct ZINC-like graphs -> supports -> tiled on the device to graphs_per_gpu graphs."""
    from gnn_matlang_amd import SpectralDesign, collate, synthetic
    from gnn_matlang_amd.graph import Batch
    pool = min(pool, graphs_per_gpu)
    raw = synthetic.make_graphs('zinc', pool, seed=seed)
    ds = SpectralDesign(recfield=2, dv=2, nfreq=7).design_many(raw)      # Zinc12k.py:12
    base = collate(ds).to(device)
    reps = (graphs_per_gpu + pool - 1) // pool
    n, B = base.x.size(0), base.num_graphs
    offs = (torch.arange(reps, device=device) * n)
    x = base.x.repeat(reps, 1)
    ei2 = (base.edge_index2.unsqueeze(1) + offs.view(1, -1, 1)).reshape(2, -1)
    ei = (base.edge_index.unsqueeze(1) + offs.view(1, -1, 1)).reshape(2, -1)
    ea = base.edge_attr2.repeat(reps, 1)
    batch = (base.batch.unsqueeze(0) + (torch.arange(reps, device=device) * B).view(-1, 1)).reshape(-1)
    ptr = (base.ptr[:-1].long().view(1, -1) + offs.view(-1, 1)).reshape(-1)
    ptr = torch.cat([ptr, torch.tensor([n * reps], device=device)])
    g = torch.Generator(device='cpu').manual_seed(seed)
    y = torch.randn(B * reps, generator=g).to(device)
    full = Batch(x=x, edge_index=ei, edge_index2=ei2, edge_attr2=ea, batch=batch, ptr=ptr.int(), y=y)
    return full, base


def cpu_baseline(host_batch, nsteps, warm, threads):
    """the oracle (port of the reference CPU algorithm) on the host cores: graphs / s"""
    from oracle import models_oracle as MO
    torch.set_num_threads(threads)
    b = host_batch
    torch.manual_seed(0)
    m = MO.zinc_gnnml3(25, 8)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    B = b.num_graphs

    def step():
        opt.zero_grad()
        l = MO.zinc_loss(m(b.x, b.edge_index2, b.edge_attr2, b.batch, B), b.y)
        l.backward()
        opt.step()
    for _ in range(warm):
        step()
    t = []
    for _ in range(nsteps):
        t0 = time.perf_counter()
        step()
        t.append(time.perf_counter() - t0)
    med = float(np.median(t))
    return dict(value=B / med, unit='graphs/s', cores=torch.get_num_threads(), kind='port',
                sample='%d ZINC-like graphs/step (N=%d, E=%d), %d timed steps after %d warm-up, median; fwd+bwd+Adam, '
                       'torch CPU fp32' % (B, b.x.size(0), b.edge_index2.size(1), nsteps, warm),
                ms_per_step=med * 1e3)


def log(*a):
    if int(os.environ.get('RANK', '0')) == 0:
        print('[bench %7.1fs]' % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=131072, help='graphs per GPU per step (SURVEY s8d: 65,536 .. 262,144 for the roofline run)')
    ap.add_argument('--pool', type=int, default=2048, help='distinct synthetic graphs (tiled to --batch)')
    ap.add_argument('--cpu-graphs', type=int, default=2048)
    ap.add_argument('--cpu-steps', type=int, default=3)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='skip the live per-kernel HIP-event timing')
    ap.add_argument('--graph', action='store_true', help='capture the whole step in a HIP graph (launch-bound small batches)')
    ap.add_argument('--ref-batch', type=int, default=64, help='also time the reference batch size (Zinc12k.py:20) as a '
                    'HIP-graph-captured step; 0 = skip')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from gnn_matlang_amd import functional as Fn, models
    from gnn_matlang_amd.dist import FlatGradSync, broadcast_parameters

    log('building data')
    data, base = build_batch(args.batch, args.pool, seed=1000 + rank, device=dev)
    log('data ready: %d graphs, %d nodes, %d support edges' % (data.num_graphs, data.x.size(0), data.edge_index2.size(1)))
    data.csr('edge_index2')                            # built once per batch (data loading, not the step)
    torch.manual_seed(0)
    model = models.zinc_gnnml3().to(dev)
    broadcast_parameters(model)
    sync = FlatGradSync(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)    # same update rule (Zinc12k.py:349), one multi-tensor kernel

    def step():
        sync.zero()
        loss = models.zinc_loss(model(data), data.y)
        loss.backward()
        sync.sync()
        opt.step()
        return loss

    def make_graph_step(mdl, dat, lr=1e-3):
        """whole train step (fwd + loss + bwd + Adam) captured once in a HIP graph, replayed per step."""
        o = torch.optim.Adam(mdl.parameters(), lr=lr, capturable=True, fused=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on a side stream (allocator, lazy init)
            for _ in range(3):
                o.zero_grad(set_to_none=True)
                models.zinc_loss(mdl(dat), dat.y).backward()
                o.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            o.zero_grad(set_to_none=True)               # gradients are handed over, not accumulated
            l_ = models.zinc_loss(mdl(dat), dat.y)
            l_.backward()
            o.step()
        return g_, l_

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    log('warm-up done')
    if not args.no_profile:
        Fn.PROFILE = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    log('timed region: %.3f s for %d steps' % (dt, args.steps))
    prof, Fn.PROFILE = Fn.PROFILE, None
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    graphs = data.num_graphs * world
    lossv = float(loss.item())
    assert np.isfinite(lossv) or os.environ.get('GML_BENCH_NOCHECK'), 'loss diverged'   # (NOCHECK: ablation builds)

    if rank == 0:
        ms = dt / args.steps * 1e3
        res = dict(metric='GNNML3 training graphs/sec on ZINC-12k', value=graphs * args.steps / dt, unit='graphs/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms, higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='Zinc12k.py GNNML3 regression train step (4x ML3Layer 30+2, S=8 supports, '
                                        'learnedge, add-pool, L1-sum, Adam 1e-3), ZINC-like synthetic graphs',
                               graphs_per_gpu=data.num_graphs, global_batch=graphs, nodes_per_gpu=int(data.x.size(0)),
                               support_edges_per_gpu=int(data.edge_index2.size(1)), supports=8,
                               parallelism='dp%d' % world, params=sum(p.numel() for p in model.parameters())),
                   final_loss=lossv)
        if prof:
            summ = Fn.profile_summary(prof)

            bf16x3 = not Fn.F32_MFMA
            BF16_PEAK_TFLOPS = 2500.0                       # dense bf16 MFMA (MI355X_MICROARCH.md)

            def roof(tag, kernel, proj_flops, edge_flops):
                """binding roof of one launch: HBM (algorithmic bytes), matrix pipe (projection flops: f32-input MFMA
                at 157.3 TF, or 3 bf16 MFMAs per product at 2.5 PF in the default bf16x3 arithmetic) or -- never
                binding here -- the fp32 VALU edge flops; achieved/peak are quoted for the binding one."""
                k = summ[tag]
                t = k['ms'] * 1e-3
                nl = k['launches']
                pf, ef = proj_flops / nl, edge_flops / nl
                gbs = k['bytes'] / t / 1e9
                t_hbm = k['bytes'] / (HBM_PEAK_GBS * 1e9)
                if bf16x3:
                    t_mat, mat_peak, mat_flops = 3 * pf / (BF16_PEAK_TFLOPS * 1e12), BF16_PEAK_TFLOPS, 3 * pf
                else:
                    t_mat, mat_peak, mat_flops = (pf + ef) / (MFMA_F32_PEAK_TFLOPS * 1e12), MFMA_F32_PEAK_TFLOPS, pf + ef
                if t_mat > t_hbm:
                    r = dict(bound='mfma', achieved=mat_flops / t / 1e12, peak=mat_peak, unit='TFLOP/s',
                             frac=mat_flops / t / 1e12 / mat_peak)
                else:
                    r = dict(bound='hbm', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s', frac=gbs / HBM_PEAK_GBS)
                r.update(traffic=None, kernel=kernel, launches=nl, avg_launch_ms=k['ms'],
                         ms_per_step=k['ms'] * nl / args.steps,
                         projection_arith='bf16x3 split on the bf16 matrix cores, fp32 accumulate' if bf16x3
                         else 'f32-input MFMA',
                         algorithmic_bytes_per_launch=k['bytes'], algorithmic_flops_per_launch=k['flops'],
                         t_hbm_ms=t_hbm * 1e3, t_matrix_ms=t_mat * 1e3, t_valu_edge_ms=ef / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3,
                         hbm_GBps=gbs, hbm_frac=gbs / HBM_PEAK_GBS,
                         algorithmic_TFLOPs=k['flops'] / t / 1e12)
                return r
            # per-step flop split of the 4 layers (Fin = 25, 32, 32, 32; Fout = 30; S = 8): projections vs edge FMAs
            N_, E_ = int(data.x.size(0)), int(data.edge_index2.size(1))
            fins = [25, 32, 32, 32]
            pj_f = sum(2 * N_ * 8 * f * 30 for f in fins) * args.steps
            ed_f = sum(2 * E_ * 8 * f for f in fins) * args.steps
            pj_b = sum(6 * N_ * 8 * f * 30 for f in fins) * args.steps          # Z, dX, dW
            ed_b = sum(4 * E_ * 8 * 30 for f in fins) * args.steps              # P update + Z.g dot
            cands = []
            if 'spectconv_bwd' in summ:
                cands.append(roof('spectconv_bwd', 'gml_k_spectconv_bwd2 / gml_k_spectconv_bwd (fused SpectConv backward: dX, dval, dW)', pj_b, ed_b))
            if 'spectconv_fwd' in summ:
                cands.append(roof('spectconv_fwd', 'gml_k_spectconv_fwd2 / gml_k_spectconv_fwd (fused SpectConv forward; the 8-wave kernel also carries the Hadamard branch)', pj_f, ed_f))
            # HBM bytes per launch from the PMC counters (collected offline with the same command under rocprofv3,
            # separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied -- profiles/r01_k_hbm_traffic.md);
            # only valid for the workload it was measured on
            tname = {32768: 'r01_k_hbm_traffic.json', 65536: 'r01_k_hbm_traffic_b65536.json',
                     131072: 'r01_k_hbm_traffic_b131072.json'}.get(data.num_graphs, 'none')
            tpath = os.path.join(ROOT, 'profiles', tname)
            if os.path.exists(tpath) and args.pool == 2048:
                tk = json.load(open(tpath))['kernels']
                for r in cands:
                    pref = 'gml_k_spectconv_bwd' if 'backward' in r['kernel'] else 'gml_k_spectconv_fwd'
                    hits = [v for kname, v in tk.items() if kname.startswith(pref)]
                    if hits:                                  # launch-weighted mean over the instantiations used
                        r['traffic'] = sum(h['hbm_bytes_per_launch'] * h.get('launches', 1) for h in hits) / \
                            sum(h.get('launches', 1) for h in hits)
                        r['traffic_source'] = 'profiles/%s (rocprofv3 PMC, per launch)' % tname
            cands.sort(key=lambda r: -r['ms_per_step'])
            res['roofline'] = cands[0]                   # the kernel with the largest share of the step
            res['roofline_other'] = cands[1:]
            res['kernels_ms_per_step'] = {tag: round(v['ms'] * v['launches'] / args.steps, 4) for tag, v in summ.items()}
        if world == 1 and not args.no_profile:
            # calibration: what torch's device-to-device copy reaches on THIS box (read + write bytes / time) -- a
            # reference point, not a ceiling (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy kernel); the
            # rooflines are quoted against the 8 TB/s specification
            ca = torch.empty(256 << 20, dtype=torch.float32, device=dev)          # 1 GiB
            cb = torch.empty_like(ca)
            for _ in range(2):
                cb.copy_(ca)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(10):
                cb.copy_(ca)
            c1.record()
            torch.cuda.synchronize()
            res['hbm_copy_GBps'] = 2 * ca.numel() * 4 * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
            del ca, cb
            # the stand-alone multi-support SpMM of the same batch (gml_spmm_fwd: H = [A_s^T X]_s materialised), the
            # bandwidth-bound piece BASELINE.json's metric names: algorithmic bytes / mean launch time (HIP events)
            csr = data.csr('edge_index2')
            S_, Fin_ = int(data.edge_attr2.size(1)), 32
            xs = torch.randn(csr.N, Fin_, device=dev)
            vals = csr.sort_values(data.edge_attr2)
            for _ in range(3):
                Fn.spmm(csr, vals, xs, S_, Fin_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nrep = 20
            e0.record()
            for _ in range(nrep):
                Fn.spmm(csr, vals, xs, S_, Fin_)
            e1.record()
            torch.cuda.synchronize()
            t_s = e0.elapsed_time(e1) / nrep * 1e-3
            q_s = 4 * (csr.E * S_ + csr.N * Fin_ + csr.N * S_ * Fin_) + 4 * (csr.E + csr.N + 1)
            res['spmm'] = {'kernel': 'gml_k_spectconv_fwd2<S, 0> via gml_spmm_fwd (8-wave SpMM, H written)', 'bound': 'hbm',
                           'achieved': q_s / t_s / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': q_s / t_s / 1e9 / HBM_PEAK_GBS,
                           'avg_launch_ms': t_s * 1e3, 'algorithmic_bytes_per_launch': q_s, 'S': S_, 'Fin': Fin_,
                           'frac_of_copy_rate': q_s / t_s / 1e9 / res['hbm_copy_GBps'],
                           'traffic_note': 'PMC at 32,768 graphs: 1045 MB / launch = 1.01x algorithmic (profiles/r01_k_spmm_hbm_traffic.md)'}
            del xs, vals
        if world == 1 and args.ref_batch > 0:
            # the reference's own batch size: launch-latency bound, so the step is replayed from a HIP graph
            rb, _ = build_batch(args.ref_batch, args.ref_batch, seed=7, device=dev)
            rb.csr('edge_index2')
            torch.manual_seed(0)
            rm = models.zinc_gnnml3().to(dev)
            gr, gl = make_graph_step(rm, rb)
            for _ in range(20):
                gr.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nrep = 200
            for _ in range(nrep):
                gr.replay()
            torch.cuda.synchronize()
            dt1 = (time.perf_counter() - t1) / nrep
            res['ref_batch'] = dict(graphs_per_step=rb.num_graphs, ms_per_step=dt1 * 1e3, value=rb.num_graphs / dt1,
                                    unit='graphs/s', mode='whole step replayed from one HIP graph',
                                    final_loss=float(gl.item()))
            log('reference batch %d: %.3f ms/step (HIP graph)' % (rb.num_graphs, dt1 * 1e3))
        if world == 1 and not args.no_cpu:
            from gnn_matlang_amd import SpectralDesign, collate, synthetic
            raw = synthetic.make_graphs('zinc', args.cpu_graphs, seed=1000)
            host = collate(SpectralDesign(recfield=2, dv=2, nfreq=7).design_many(raw))
            # ATen intra-op threading of these small index ops stops scaling long before this box's core
            # count; time a short ladder of thread counts and report the best one (cores = threads used)
            best = None
            for th in [t for t in (8, 16, 32, 64) if t <= (os.cpu_count() or 1)] or [os.cpu_count() or 1]:
                log('cpu baseline: %d graphs on %d threads' % (host.num_graphs, th))
                r = cpu_baseline(host, args.cpu_steps, 1, th)
                log('   -> %.0f graphs/s' % r['value'])
                if best is None or r['value'] > best['value']:
                    best = r
            best['host_cores'] = os.cpu_count()
            res['cpu_baseline'] = best
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
