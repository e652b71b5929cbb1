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
