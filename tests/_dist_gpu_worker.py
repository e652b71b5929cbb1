"""Rank program of tests/test_gpu_dist.py: world_size ranks that SHARE cuda:0 (the GPU box has one device), gloo
backend with CUDA tensors.  Drives the PRODUCT model (gnn_matlang_amd.models.zinc_gnnml3 -> libgml_hip.so) through the
data-parallel step: graphs sharded over ranks, FlatGradSync (p.grad handed over as views of the flat buffer), fused
Adam.  Rank 0 writes the first step's gradients and the 3-step loss trajectory."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = lambda a: torch.tensor(np.asarray(a))


def main():
    out, balanced = sys.argv[1], sys.argv[2] == '1'
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    # a device per rank: RCCL ('nccl'), as on the 8-GPU node; the 1-GPU test box: both ranks on cuda:0 over gloo
    nccl = torch.cuda.device_count() >= world and os.environ.get('GML_TEST_GLOO') != '1'
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', rank)) if nccl else 0)
    torch.cuda.set_device(dev)
    if nccl:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from gnn_matlang_amd import models
    from gnn_matlang_amd.dist import FlatGradSync, broadcast_parameters
    from gnn_matlang_amd.graph import Batch, shard_graphs, shard_graphs_balanced
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'model_zinc_gnnml3.npz'))
    b = {k[len('batch/'):]: g[k] for k in g.files if k.startswith('batch/')}
    B = len(b['y'])
    if balanced:                                           # split by support edges per graph
        gid = b['batch'][b['edge_index2'][1]]
        lo, hi = shard_graphs_balanced(np.bincount(gid, minlength=B), rank, world)
    else:
        lo, hi = shard_graphs(B, rank, world)
    nodes = np.flatnonzero((b['batch'] >= lo) & (b['batch'] < hi))
    n0, n1 = nodes[0], nodes[-1] + 1
    em = (b['edge_index2'][1] >= n0) & (b['edge_index2'][1] < n1)
    e1 = (b['edge_index'][1] >= n0) & (b['edge_index'][1] < n1)
    bt = b['batch'][n0:n1] - lo
    ptr = np.concatenate([[0], np.cumsum(np.bincount(bt, minlength=hi - lo))]).astype(np.int32)
    data = Batch(x=T(b['x'][n0:n1]), edge_index=T(b['edge_index'][:, e1] - n0), edge_index2=T(b['edge_index2'][:, em] - n0),
                 edge_attr2=T(b['edge_attr2'][em]), batch=T(bt), ptr=T(ptr), y=T(b['y'][lo:hi])).to(dev)
    torch.manual_seed(100 + rank)                          # replicas start different ...
    m = models.zinc_gnnml3()
    if rank == 0:
        m.load_state_dict({k[len('param/'):]: T(g[k]) for k in g.files if k.startswith('param/')})
    m = m.to(dev)
    broadcast_parameters(m)                                # ... and are made identical
    sync = FlatGradSync(m.parameters())
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    losses, grads = [], None
    for step in range(3):
        sync.zero()
        l = models.zinc_loss(m(data), data.y)
        l.backward()
        flat = sync.sync()
        if step == 0:
            assert flat is not None and all(p.grad.data_ptr() >= flat.data_ptr() for p in sync.params)   # views of the flat buffer
            grads = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
        opt.step()
        lt = l.detach().clone()
        dist.all_reduce(lt)
        losses.append(float(lt.item()))
    if rank == 0:
        torch.save(dict(grads=grads, losses=losses, shards=[lo, hi], world=dist.get_world_size()), out)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
