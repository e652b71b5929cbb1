"""CPU, world_size 2, gloo: the data-parallel step (graphs sharded over ranks, ONE sum all-reduce of the
flat gradient buffer) reproduces the single-process gradient of the same global batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN

T = lambda a: torch.tensor(np.asarray(a))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(b, lo, hi):
    """graphs [lo, hi) of a collated batch dict (block diagonal => just slice and re-base)."""
    nodes = np.flatnonzero((b['batch'] >= lo) & (b['batch'] < hi))
    n0, n1 = nodes[0], nodes[-1] + 1
    em = (b['edge_index2'][1] >= n0) & (b['edge_index2'][1] < n1)
    return dict(x=b['x'][n0:n1], edge_index2=b['edge_index2'][:, em] - n0, edge_attr2=b['edge_attr2'][em],
                batch=b['batch'][n0:n1] - lo, y=b['y'][lo:hi])


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gnn_matlang_amd.dist import FlatGradSync, broadcast_parameters
    from gnn_matlang_amd.graph import shard_graphs
    from oracle import models_oracle as MO
    torch.set_num_threads(1)
    g = np.load(os.path.join(GOLDEN, 'model_zinc_gnnml3.npz'))
    b = {k[len('batch/'):]: g[k] for k in g.files if k.startswith('batch/')}
    torch.manual_seed(100 + rank)                      # replicas start different ...
    m = MO.zinc_gnnml3(25, 8)
    if rank == 0:
        m.load_state_dict({k[len('param/'):]: T(g[k]) for k in g.files if k.startswith('param/')})
    broadcast_parameters(m)                            # ... and are made identical
    sync = FlatGradSync(m.parameters())
    lo, hi = shard_graphs(len(b['y']), rank, world)
    s = _shard(b, lo, hi)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for step in range(3):
        sync.zero()
        pre = m(T(s['x']), T(s['edge_index2']), T(s['edge_attr2']), T(s['batch']), hi - lo)
        l = MO.zinc_loss(pre, T(s['y']))
        l.backward()
        if step == 0:
            sync.sync()
            grads = {n: p.grad.clone() for n, p in m.named_parameters()}
        else:
            sync.sync()
        opt.step()
        lt = l.detach().clone()
        dist.all_reduce(lt)
        losses.append(lt.item())
    if rank == 0:
        torch.save(dict(grads=grads, losses=losses), out)
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process(tmp_path):
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    g = np.load(os.path.join(GOLDEN, 'model_zinc_gnnml3.npz'))
    for n, v in r['grads'].items():                    # == the reference's full-batch gradients
        np.testing.assert_allclose(v.numpy(), g['grad/' + n], rtol=1e-4, atol=2e-5, err_msg=n)
    np.testing.assert_allclose(r['losses'], g['loss_traj'][:3], rtol=1e-4)


def _worker_mnist(rank, world, port, out):
    """config 4 (the one BASELINE shards over GPUs) with its readout batch norm: statistics all-reduced (dist.SyncBatchNorm1d)"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gnn_matlang_amd.dist import FlatGradSync, SyncBatchNorm1d
    from gnn_matlang_amd.graph import shard_graphs
    from oracle import models_oracle as MO
    torch.set_num_threads(1)
    g = np.load(os.path.join(GOLDEN, 'model_mnist_gnnml3_tf.npz'))
    b = {k[len('batch/'):]: g[k] for k in g.files if k.startswith('batch/')}
    B = len(b['y'])
    m = MO.mnist_gnnml3().train()
    m.bnr = SyncBatchNorm1d(m.bnr.num_features, eps=m.bnr.eps, momentum=m.bnr.momentum)     # same keys as BatchNorm1d
    m.load_state_dict({k[len('param/'):]: T(g[k]) for k in g.files if k.startswith('param/')}, strict=False)
    sync = FlatGradSync(m.parameters())
    lo, hi = shard_graphs(B, rank, world)
    s = _shard(b, lo, hi)
    sync.zero()
    pre = m(T(s['x']), T(s['edge_index2']), T(s['edge_attr2']), T(s['batch']), hi - lo)
    # the reference's loss is the batch MEAN of the cross entropy (libs/metrics_tf.py): sum over the shard / GLOBAL batch size
    l = torch.nn.functional.cross_entropy(pre, T(s['y']).long(), reduction='sum') / B
    l.backward()
    sync.sync()
    lt = l.detach().clone()
    dist.all_reduce(lt)
    logits = [torch.zeros(hi - lo if r == rank else shard_graphs(B, r, world)[1] - shard_graphs(B, r, world)[0], pre.size(1)) for r in range(world)]
    dist.all_gather(logits, pre.detach()) if len({t.size(0) for t in logits}) == 1 else None
    if rank == 0:
        torch.save(dict(grads={n: p.grad.clone() for n, p in m.named_parameters()}, loss=lt.item(),
                        logits=torch.cat(logits) if len({t.size(0) for t in logits}) == 1 else None,
                        running_mean=m.bnr.running_mean.clone()), out)
    dist.destroy_process_group()


def test_two_rank_mnist_step_with_synced_readout_batchnorm(tmp_path):
    out = str(tmp_path / 'm0.pt')
    mp.spawn(_worker_mnist, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    g = np.load(os.path.join(GOLDEN, 'model_mnist_gnnml3_tf.npz'))
    np.testing.assert_allclose(r['loss'], float(g['loss']), rtol=1e-5)
    if r['logits'] is not None:
        np.testing.assert_allclose(r['logits'].numpy(), g['logits'], rtol=1e-4, atol=2e-5)
    for n, v in r['grads'].items():                    # == the TF graph's full-batch gradients
        ref = g['grad/' + n]
        assert np.abs(v.numpy() - ref).max() <= 1e-4 * np.abs(ref).max(), n


def _worker_trainstep(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gnn_matlang_amd.dist import TrainStep, broadcast_parameters
    from gnn_matlang_amd.graph import shard_graphs
    from oracle import models_oracle as MO
    torch.set_num_threads(1)
    g = np.load(os.path.join(GOLDEN, 'model_zinc_gnnml3.npz'))
    b = {k[len('batch/'):]: g[k] for k in g.files if k.startswith('batch/')}
    torch.manual_seed(7 + rank)
    m = MO.zinc_gnnml3(25, 8)
    if rank == 0:
        m.load_state_dict({k[len('param/'):]: T(g[k]) for k in g.files if k.startswith('param/')})
    broadcast_parameters(m)
    lo, hi = shard_graphs(len(b['y']), rank, world)
    s = _shard(b, lo, hi)

    class D(object):
        pass
    d = D()
    d.x, d.ei, d.ea, d.batch, d.y, d.B = T(s['x']), T(s['edge_index2']), T(s['edge_attr2']), T(s['batch']), T(s['y']), hi - lo
    ts = TrainStep(m, lambda mod, dd: MO.zinc_loss(mod(dd.x, dd.ei, dd.ea, dd.batch, dd.B), dd.y), torch.optim.Adam(m.parameters(), lr=1e-3))
    losses, order = [], None
    for step in range(3):
        l = ts(d)
        if step == 0:
            order = list(ts.order)
            grads = {n: p.grad.clone() for n, p in m.named_parameters()}
            # the all-reduced gradients are views of ONE flat buffer: what the optimizer reads is what the collective wrote
            flat = ts.sync.flat
            assert all(flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4 for p in ts.sync.params)
        lt = l.clone()
        dist.all_reduce(lt)
        losses.append(lt.item())
    if rank == 0:
        torch.save(dict(grads=grads, losses=losses, order=order), out)
    dist.destroy_process_group()


def test_trainstep_orders_backward_fold_allreduce_optimizer(tmp_path):
    """dist.TrainStep -- the sequence bench.py times and, on RCCL, captures into one HIP graph with the all-reduce inside: zero ->
    forward -> backward (folds flushed) -> ONE flat all-reduce -> optimizer.  Two gloo ranks on the oracle model: the order is as
    stated, the summed shard gradients are the reference's full-batch gradients and the loss trajectory is the fixture's."""
    out = str(tmp_path / 'ts.pt')
    mp.spawn(_worker_trainstep, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res['order'] == ['zero', 'forward', 'backward', 'allreduce', 'optimizer']
    g = np.load(os.path.join(GOLDEN, 'model_zinc_gnnml3.npz'))
    for n, v in res['grads'].items():
        ref = g['grad/' + n]
        assert np.abs(v.numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30), n
    np.testing.assert_allclose(res['losses'], g['loss_traj'][:3], rtol=1e-5)
