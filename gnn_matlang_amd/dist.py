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


class _SyncBNFunction(torch.autograd.Function):
    """BatchNorm over the GLOBAL batch of a data-parallel step: one all-reduce of [sum x, sum x^2, count] (2 C + 1 floats)
    forward, one of [sum dy, sum dy xhat] backward.  With these the R-rank step equals the 1-rank step on the same global
    batch (config 4: the MNIST readout's tf.layers.batch_normalization, libs/layers_tf.py:343-349)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        C = x.size(1)
        st = torch.cat([x.sum(0), (x * x).sum(0), x.new_tensor([float(x.size(0))])])
        dist.all_reduce(st, op=dist.ReduceOp.SUM, group=group)
        n = st[-1]
        mean = st[:C] / n
        var = (st[C:2 * C] / n - mean * mean).clamp_(min=0)            # biased, as batch norm normalises with
        rstd = torch.rsqrt(var + eps)
        xhat = (x - mean) * rstd
        ctx.save_for_backward(xhat, weight, rstd, n)
        ctx.group, ctx.has_bias = group, bias is not None
        ctx.mark_non_differentiable(mean, var, n)
        y = xhat if weight is None else xhat * weight            # affine=False: weight and bias are None
        return (y if bias is None else y + bias), mean, var, n

    @staticmethod
    def backward(ctx, dy, _dm, _dv, _dn):
        xhat, weight, rstd, n = ctx.saved_tensors
        C = dy.size(1)
        sums = torch.cat([dy.sum(0), (dy * xhat).sum(0)])            # local sums = the parameter gradients of this shard
        g = sums.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        dx = (dy - g[:C] / n - xhat * (g[C:] / n)) * (rstd if weight is None else weight * rstd)
        return dx, (sums[C:] if weight is not None else None), (sums[:C] if ctx.has_bias else None), None, None


class SyncBatchNorm1d(torch.nn.BatchNorm1d):
    """torch.nn.BatchNorm1d (same parameters, buffers and state_dict keys) whose TRAINING statistics are those of the global
    batch when a process group with more than one rank is initialised; with one rank, or in eval mode, it is the base class."""

    def forward(self, x):
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # batch statistics are used in training mode AND whenever no running statistics exist (track_running_stats=False):
        # both take the synchronised road, so that the R-rank step equals the 1-rank step.  Every rank must run
        # the layer in the same mode -- the all-reduces are collective.
        use_batch_stats = self.training or self.running_mean is None
        if not use_batch_stats or world == 1 or x.dim() != 2:
            return super().forward(x)
        y, mean, var, n = _SyncBNFunction.apply(x, self.weight, self.bias, self.eps, None)
        if not self.training or self.running_mean is None:
            return y
        with torch.no_grad():
            self.num_batches_tracked += 1
            m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
            self.running_mean.mul_(1 - m).add_(mean, alpha=m)
            self.running_var.mul_(1 - m).add_(var * (n / (n - 1).clamp(min=1)), alpha=m)
        return y


class TrainStep(object):
    """One data-parallel training step in the order its pieces need:

        zero -> forward + loss -> backward (the weight-gradient folds deferred: functional.deferred_folds) -> fold flush
             -> ONE flat SUM all-reduce (FlatGradSync.sync; nothing with one rank) -> optimizer

    ``step(data)`` runs it eagerly.  ``capture(data)`` records the SAME sequence, the all-reduce included, into one HIP graph
    over static tensors and returns a replay function (RCCL collectives are stream-ordered and capturable like kernels; the flat
    gradient buffer and every .grad view live in the graph's private pool, so a replay rewrites the memory the optimizer reads).
    ``loss_fn(model, data) -> scalar``.  SURVEY s8e: the collective is latency-bound at the reference's batch size -- inside the graph
    it costs no host launch."""

    def __init__(self, model, loss_fn, optimizer, sync=None, defer_folds=True, group=None):
        self.model, self.loss_fn, self.opt = model, loss_fn, optimizer
        self.sync = sync if sync is not None else FlatGradSync(model.parameters(), group)
        self.defer = bool(defer_folds)
        self.order = []                                        # what ran, in order (tests)

    def step(self, data):
        self.order = ['zero']
        self.sync.zero()
        loss = self.loss_fn(self.model, data)
        self.order.append('forward')
        if self.defer and loss.is_cuda:
            from . import functional as Fn
            with Fn.deferred_folds(list(self.model.parameters())):      # inert (plain folds) if a .grad survived zero()
                loss.backward()
            self.order.append('backward+fold')
        else:
            loss.backward()
            self.order.append('backward')
        flat = self.sync.sync()
        self.order.append('allreduce' if flat is not None else 'no-allreduce')
        self.opt.step()
        self.order.append('optimizer')
        return loss.detach()

    __call__ = step

    def capture(self, data, warmup=3):
        """-> (replay, loss tensor): ``replay()`` re-runs the captured step on the CURRENT contents of data's tensors."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                          # warm-up off the capture stream (allocator, lazy init, RCCL channels)
            for _ in range(warmup):
                self.step(data)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = self.step(data)
        return g.replay, loss
