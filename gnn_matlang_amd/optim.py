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
    Parameters without a gradient at the FIRST step are left out for good (the job list is built once per gradient layout).

    Checkpoints: ``state[p]`` holds ``step`` (the device count, one tensor shared by the parameters of a launch), ``exp_avg`` and
    ``exp_avg_sq`` -- torch.optim.Adam's keys -- so ``state_dict()`` / ``load_state_dict()`` round-trip the count and the moments
    (loading re-packs them into the flat buffer at the next step).  ``lr`` / ``betas`` / ``eps`` are passed to the launch BY VALUE:
    a captured step (HIP graph) replays the values it was captured with -- a scheduler changing ``param_groups[..]['lr']`` needs a
    re-capture (eager steps read the current value every time)."""

    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._built = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._built = None                                   # the loaded moments / count are re-packed by the next step

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
                for row, key in enumerate(('exp_avg', 'exp_avg_sq')):
                    view = flat[row, off:off + p.numel()].view_as(p)
                    if key in st:                            # loaded from a checkpoint (or a rebuild): keep the values
                        view.copy_(st[key])
                    st[key] = view
                off += p.numel()
            for i in range(0, len(ps), _lib.GML_ADAM_MAX_JOBS):
                part = ps[i:i + _lib.GML_ADAM_MAX_JOBS]
                # (every chunk advances a step count of its own, in lockstep: no copy between the launches of one optimizer step)
                step = torch.zeros(1, dtype=torch.float32, device=dev)
                old = [self.state[p]['step'] for p in part if 'step' in self.state[p]]
                if old:                                      # resume: the count the checkpoint holds (the same for every parameter)
                    step.fill_(float(torch.as_tensor(old[0]).reshape(-1)[0]))
                for p in part:
                    self.state[p]['step'] = step
                chunks.append(dict(group=group, ps=part, step=step, done=torch.zeros(1, dtype=torch.int32, device=dev), dev=dev))
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
