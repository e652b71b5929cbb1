"""Data parallelism over graphs: one process per GPU, every rank holds a full model
replica and its own shard of each global batch, and ONE sum all-reduce of a flat fp32 gradient
buffer per step keeps the replicas identical (RCCL over xGMI on the GPU box: backend 'nccl';
gloo in the CPU tests).  SUM, not mean: the reference losses are reduction='sum' and never divided
by the batch size (Zinc12k.py:365, counting.py:411), so the summed shard gradients equal the
single-GPU gradient of the same global batch.  The layer itself has no cross-graph exchange, so no
other collective exists on the path."""
import torch
import torch.distributed as dist


class FlatGradSync(object):
    """One all-reduce per step over a flat fp32 gradient buffer.

    ``zero()`` drops the gradients (``.grad = None``), so the backward pass hands each parameter's freshly
    written gradient tensor over without an accumulate kernel per parameter; ``sync()`` packs them into one
    flat buffer (a single concatenation), all-reduces it and gives every parameter its slice back as ``.grad``
    (views, no copies).  With one rank nothing is launched at all."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.flat = None

    def _world(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def zero(self):
        for p in self.params:
            p.grad = None

    def sync(self):
        if self._world() == 1:
            return None
        self.flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                               for p in self.params])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        return self.flat


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank ``src``'s weights (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    ps = [p.data for p in module.parameters()] + [b.data for b in module.buffers() if b.dtype.is_floating_point]
    flat = torch.cat([p.reshape(-1) for p in ps])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for p in ps:
        p.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
