"""Optimiser update rules of the reference's two front ends.

The PyTorch scripts use ``torch.optim.Adam`` (Zinc12k.py:349 ...): take torch's.  The TensorFlow pipeline of config 4 uses
``tf.train.AdamOptimizer`` (libs/models_tf.py:201), whose update places epsilon differently:

    torch   p -= lr / (1 - b1^t) * m / ( sqrt(v) / sqrt(1 - b2^t) + eps )
    TF      p -= lr * sqrt(1 - b2^t) / (1 - b1^t) * m / ( sqrt(v) + eps )          (eps "hat" of the paper)

i.e. TF's epsilon is torch's scaled by 1 / sqrt(1 - b2^t) (31.6 at the first step).  With lr = 0.01 and many parameters whose
first gradients are ~1e-7 (dead relu units of the first layer) the two rules move those parameters by different
amounts -- a 2 % difference of the loss after five steps on the MNIST-75 fixture -- so parity with the TF reference needs
TF's rule.  ``TFAdam`` is that rule (multi-tensor ``_foreach`` ops: a handful of launches per step on the GPU)."""
import math

import torch


class TFAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            b1, b2 = group['betas']
            for p in ps:
                st = self.state[p]
                if not st:
                    st['step'], st['m'], st['v'] = 0, torch.zeros_like(p), torch.zeros_like(p)
            t = self.state[ps[0]]['step'] + 1
            for p in ps:
                self.state[p]['step'] = t
            gs, ms, vs = [p.grad for p in ps], [self.state[p]['m'] for p in ps], [self.state[p]['v'] for p in ps]
            torch._foreach_mul_(ms, b1)
            torch._foreach_add_(ms, gs, alpha=1 - b1)
            torch._foreach_mul_(vs, b2)
            torch._foreach_addcmul_(vs, gs, gs, value=1 - b2)
            den = torch._foreach_sqrt(vs)
            torch._foreach_add_(den, group['eps'])
            lr_t = group['lr'] * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            torch._foreach_addcdiv_(ps, ms, den, value=-lr_t)


class OneLaunchAdam(torch.optim.Optimizer):
    """torch.optim.Adam's update (no weight decay, no amsgrad: Zinc12k.py:349) as ONE launch over all parameter tensors
    (gml_adam_many, csrc/gml_misc.hip) -- for the reference's batch size, where a training step is a chain of ~30 launches of
    microseconds each and torch's fused Adam is three of them.  The step count lives on the device (capturable: every replay of a
    captured step advances it).  Moments are kept in one flat buffer; gradients may be any tensors (views of the fold buffers).
    Parameters without a gradient at the FIRST step are left out for good (the job list is built once per gradient layout)."""

    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._built = None

    def _build(self):
        from . import _lib
        import ctypes
        chunks = []
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps):
                raise ValueError('OneLaunchAdam: contiguous float32 parameters on the GPU')
            flat = torch.zeros(2, sum(p.numel() for p in ps), dtype=torch.float32, device=dev)
            off = 0
            for p in ps:
                st = self.state[p]
                st['exp_avg'], st['exp_avg_sq'] = flat[0, off:off + p.numel()].view_as(p), flat[1, off:off + p.numel()].view_as(p)
                off += p.numel()
            for i in range(0, len(ps), _lib.GML_ADAM_MAX_JOBS):
                part = ps[i:i + _lib.GML_ADAM_MAX_JOBS]
                # (every chunk advances a step count of its own, in lockstep: no copy between the launches of one optimizer step)
                chunks.append(dict(group=group, ps=part, step=torch.zeros(1, dtype=torch.float32, device=dev),
                                   done=torch.zeros(1, dtype=torch.int32, device=dev), dev=dev))
        self._built = chunks

    @torch.no_grad()
    def step(self):
        from . import _lib
        from .graph import _ptr, _stream
        import ctypes
        if self._built is None:
            self._build()
        for c in self._built:
            ps = c['ps']
            arr = (_lib.AdamJob * len(ps))()
            for k, p in enumerate(ps):
                g = p.grad
                if g is None:
                    raise RuntimeError('OneLaunchAdam: a parameter that had a gradient at the first step has none now')
                if not g.is_contiguous():
                    g = g.contiguous()
                st = self.state[p]
                arr[k].p, arr[k].g, arr[k].m, arr[k].v, arr[k].n = p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel()
            b1, b2 = c['group']['betas']
            with torch.cuda.device(c['dev']):
                _lib.call('gml_adam_many', ctypes.addressof(arr), len(ps), _ptr(c['step']), _ptr(c['done']), float(c['group']['lr']), float(b1),
                          float(b2), float(c['group']['eps']), _stream(c['dev']))
