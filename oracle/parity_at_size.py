"""Parity of one ZINC GNNML3 train step AT BENCH SIZE against the oracle in float64 (TEST INFRASTRUCTURE, see oracle/__init__.py;
used by tests/test_gpu_parity.py and by the checker leg of bench.py -- never by the package).

The bench's batch (bench.build_batch) tiles a pool of P distinct graphs R times and gives every copy its own target y.  Graphs
are independent (block-diagonal batch, per-graph pooling; Zinc12k.py:338-343), so in exact arithmetic

    logit(copy r of graph g)            = pre_g                                   (the same for every copy)
    d loss / d theta over the big batch = sum_g c_g d pre_g / d theta,   c_g = sum_r sign(pre_g - y[r, g])      (L1-sum loss, :365)

and ONE float64 forward + backward of the oracle over the P pool graphs is the exact reference for all B = P R graphs.

The criterion for a parameter gradient is the term-sum one: |got - ref| <= tol * T, T = the sum of the
absolute values of the TERMS the element is a sum of -- at the level of the products the kernels actually form (a conv weight
gradient is X^T (A_s G): 3 M rows x ~6 edges of |x| |val| |g| products per element), per ML3Layer by oracle/termsums.py from
the float64 activations at the layer's input and the float64 gradient at its output (layer-local: see model_termsums); every
copy of a graph contributes the same magnitudes, so T(batch) = R * T(pool).  The head's Linear layers: T_W = |g|^T |input|, T_b = sum |g|.
(Round 5 first tried T from whole-graph contributions -- one float64 backward pass per pool graph: slow, and blind to the
cancellation INSIDE a graph, which is where most of a conv weight gradient's terms cancel: the elements that failed under it
sit at 2e-5 of their tensor's maximum.)
"""
import time

import numpy as np
import torch

from . import models_oracle as MO
from .termsums import ml3_termsums


def reference(pool_batch, state_dict, y_full, pre_dev=None, T=None, threads=16, head_pre_dev=None, layer_masks_dev=None):
    """pool_batch: the collated pool (CPU tensors x, edge_index2, edge_attr2, batch, num_graphs); state_dict: the model's
    parameters (any device); y_full [R * P] targets in the bench's order (copy-major); pre_dev [R * P]: the device's own logits -- the
    L1 loss's sign(pre - y) is then taken from THEM (a copy whose |pre - y| is below the logits' round-off would otherwise flip the
    sign of its whole contribution: a property of the loss, not an error of the backward); head_pre_dev [R * P, nh]: the device's
    fc1 outputs -- the head's relu mask is then taken from them for the same reason (a hidden unit of the head whose pre-activation is
    within round-off of zero switches a whole graph's contribution, coherently in all R copies: with 2,048 pool graphs ONE such unit is
    5e-4 of fc1.weight's term sum; the relus inside the layers act per node, 1 / 3 M of a sum, and need no such care); T: term sums of
    an earlier call with the same parameters and pool (they do not depend on the arithmetic mode); layer_masks_dev {layer index: bool
    [>= n_pool, nout1]}: the device's activation pattern of that layer's relu(conv) columns (first copy: the copies share their values on
    the device as in exact arithmetic -- every row's result depends on its own row only) -- the float64 pass is then evaluated WITH that
    pattern (a = z * mask): "given the forward's discrete decisions, are the gradients right", the same treatment as the loss's sign and
    the head's mask.  Returns dict(pre [P] float64,
    grads {name: float64 array}, T {name: float64 array}, head_units_flipped, seconds)."""
    t0 = time.perf_counter()
    torch.set_num_threads(min(int(threads), torch.get_num_threads()))   # (tiny float64 ops: hundreds of threads only contend)
    b = pool_batch
    P = int(b.num_graphs)
    R = int(y_full.numel()) // P
    m = MO.zinc_gnnml3(int(b.x.size(1)), int(b.edge_attr2.size(1))).double()
    m.load_state_dict({k: v.detach().cpu().double() for k, v in state_dict.items()})
    head = {}
    hook = m.fc1.register_forward_hook(lambda mod, inp, out: head.__setitem__('z1', out))
    n0 = int(b.x.size(0))
    for i, mk in (layer_masks_dev or {}).items():
        lay = getattr(m, 'conv%d' % i)
        lay._relu_mask = mk[:n0, :int(lay.conv1.weight.size(2))].detach().cpu()
    pre = m(b.x.double(), b.edge_index2, b.edge_attr2.double(), b.batch, P)[:, 0]            # [P]
    hook.remove()
    y = y_full.detach().cpu().double().view(R, P)
    pd = pre.detach().unsqueeze(0) if pre_dev is None else pre_dev.detach().cpu().double().view(R, P)
    c = torch.sign(pd - y)                                                                    # [R, P]
    params = dict(m.named_parameters())
    names = list(params)
    flipped = 0
    if head_pre_dev is None:
        g = torch.autograd.grad((c.sum(0) * pre).sum(), [params[n] for n in names], retain_graph=True)
        grads = {n: v.numpy() for n, v in zip(names, g)}
    else:
        z1 = head['z1']                                                                       # [P, nh], in the graph (Zinc12k.py:343-345)
        mk = (head_pre_dev.detach().cpu().view(R, P, -1) > 0).double()
        flipped = int((mk != (z1.detach() > 0).double().unsqueeze(0)).sum())
        gz = (c.unsqueeze(-1) * mk).sum(0) * m.fc2.weight.detach()                           # d loss / d z1, summed over the copies
        low = [n for n in names if not n.startswith('fc2.')]
        g = torch.autograd.grad(z1, [params[n] for n in low], grad_outputs=gz, retain_graph=True)
        grads = {n: v.numpy() for n, v in zip(low, g)}
        cg = c.sum(0)
        grads['fc2.weight'] = (cg.unsqueeze(1) * torch.relu(z1.detach())).sum(0, keepdim=True).numpy()
        grads['fc2.bias'] = cg.sum().reshape(1).numpy()
    if T is None:
        T = model_termsums(m, b, P, R)
    return dict(pre=pre.detach().numpy(), grads=grads, T=T, R=R, P=P, head_units_flipped=flipped, seconds=time.perf_counter() - t0)


def model_termsums(m, b, P, R):
    """Term sums of every parameter gradient of the float64 oracle model `m` over the pool batch `b`, for the UNSIGNED loss
    sum_g pre_g (the copies' signs only flip terms), times R copies.  LAYER-LOCAL: every layer's input activations and the
    gradient at its output are taken from the float64 pass as data, and T is the sum of the |x| |val| |g| |w| products that layer's
    kernels form (oracle/termsums.py); the head's Linear layers: T_W = |g|^T |input|, T_b = sum |g|.  Typical T / |gradient|: 10 - 1000.
    (A form that carries term sums THROUGH the layers in a forward and a backward sweep -- the running-error bound -- was tried: it
    compounds to T / |gradient| ~ 1e9 over four layers and would accept anything; not used.)  What the local form does not absorb is
    round-off that ARRIVES from the layers below: an activation that is right to 1e-5 of its own term sum but is a cancelling sum
    (T_x / |x| ~ 30) enters the next layer's products with 3e-4 relative error, and that counts against the upper layer here."""
    acts, hooks = {}, []
    for i in range(1, m.nlayers + 1):
        lay = getattr(m, 'conv%d' % i)
        hooks.append(lay.register_forward_hook(lambda mod, inp, out, i=i: acts.__setitem__(i, (inp[0], out))))
    hooks.append(m.fc1.register_forward_hook(lambda mod, inp, out: acts.__setitem__('fc1', (inp[0], out))))
    hooks.append(m.fc2.register_forward_hook(lambda mod, inp, out: acts.__setitem__('fc2', (inp[0], out))))
    pre1 = m(b.x.double(), b.edge_index2, b.edge_attr2.double(), b.batch, P)[:, 0]
    for h in hooks:
        h.remove()
    keys = list(acts)
    gouts = dict(zip(keys, torch.autograd.grad(pre1.sum(), [acts[k][1] for k in keys])))
    ea64, ei = b.edge_attr2.double(), b.edge_index2
    S = int(ea64.size(1))
    T = {}
    for k in range(1, m.nlayers + 1):
        lay = getattr(m, 'conv%d' % k)
        params = {n: v.detach() for n, v in lay.named_parameters()}
        nout2 = int(params['fc11.weight'].size(0)) if 'fc11.weight' in params else 0
        meta = (1, S, S, int(acts[k][0].size(1)), int(params['conv1.weight'].size(2)), nout2)
        ts = ml3_termsums(meta, acts[k][0].detach(), ea64, ei, gouts[k].detach(), params)
        for n in params:
            T['conv%d.%s' % (k, n)] = ts[n] * float(R)
    for name in ('fc1', 'fc2'):
        g = gouts[name].detach().abs()
        T[name + '.weight'] = (g.t() @ acts[name][0].detach().abs()).numpy() * float(R)
        T[name + '.bias'] = g.sum(0).numpy() * float(R)
    return T


def oracle_fp32_as_device(pool_batch, state_dict, y_full, pre_dev, head_pre_dev=None, threads=16):
    """The reference arithmetic itself as the thing under test: the oracle in float32 on the CPU over the pool, its copies weighted by
    the same signs as the device's run (pre_dev) and -- like the device's run in reference() -- its own head mask.  Returns
    (pre [R * P] float32 logits tiled, {name: gradient}): what compare() takes.  Shows what the criterion says about plain fp32
    (a relu unit of a layer within round-off of zero flips in fp32 as it does on the device)."""
    torch.set_num_threads(min(int(threads), torch.get_num_threads()))
    b = pool_batch
    P = int(b.num_graphs)
    R = int(y_full.numel()) // P
    m = MO.zinc_gnnml3(int(b.x.size(1)), int(b.edge_attr2.size(1))).float()
    m.load_state_dict({k: v.detach().cpu().float() for k, v in state_dict.items()})
    head = {}
    hook = m.fc1.register_forward_hook(lambda mod, inp, out: head.__setitem__('z1', out.detach()))
    pre = m(b.x.float(), b.edge_index2, b.edge_attr2.float(), b.batch, P)[:, 0]
    hook.remove()
    y = y_full.detach().cpu().float().view(R, P)
    c = torch.sign(pre_dev.detach().cpu().float().view(R, P) - y).sum(0)
    params = dict(m.named_parameters())
    g = torch.autograd.grad((c * pre).sum(), list(params.values()))
    return pre.detach().repeat(R).numpy(), {n: v.numpy() for n, v in zip(params, g)}, head['z1'].repeat(R, 1)


def relu_mask_diffs(pool_batch, state_dict, y_full, pre_dev, outs_dev):
    """How many relu units of each ML3Layer's conv columns come out on the OTHER side of zero on the device than in float64 (first
    copy of the pool: outs_dev = {layer index: [>= n_pool, C] device output}), and how much of a column's gradient mass sits on such
    units.  A unit within the forward round-off of zero flips its derivative -- in every copy together, since the copies share their
    values -- and on a column with few live nodes ONE such node is 1e-3 .. 1e-2 of the column's term sum: the one way the bench-size
    criterion is exceeded in a trained state, in either arithmetic (it is a property of relu at zero, not an error of a kernel)."""
    b = pool_batch
    P = int(b.num_graphs)
    R = int(y_full.numel()) // P
    m = MO.zinc_gnnml3(int(b.x.size(1)), int(b.edge_attr2.size(1))).double()
    m.load_state_dict({k: v.detach().cpu().double() for k, v in state_dict.items()})
    live = {}
    hooks = [getattr(m, 'conv%d' % i).register_forward_hook(lambda mod, inp, out, i=i: live.__setitem__(i, out)) for i in range(1, m.nlayers + 1)]
    pre = m(b.x.double(), b.edge_index2, b.edge_attr2.double(), b.batch, P)[:, 0]
    for h in hooks:
        h.remove()
    c = torch.sign(pre_dev.detach().cpu().double().view(R, P) - y_full.detach().cpu().double().view(R, P)).sum(0)
    gouts = torch.autograd.grad((c * pre).sum(), [live[i] for i in sorted(live)])
    n0 = int(b.x.size(0))
    rep = {}
    for (i, g) in zip(sorted(live), gouts):
        if i not in outs_dev:
            continue
        c1 = int(getattr(m, 'conv%d' % i).conv1.weight.size(2))
        ref_on = live[i].detach()[:, :c1] > 0
        dev_on = outs_dev[i][:n0, :c1].detach().cpu() > 0
        diff = ref_on != dev_on
        ga = g[:, :c1].abs()
        col = (ga * diff).sum(0) / (ga * ref_on).sum(0).clamp(min=1e-300)
        rep['layer%d' % i] = dict(units=int(diff.numel()), differing=int(diff.sum()), worst_column_share_of_gradient_mass=float(col.max()))
    return rep


def compare(ref, pre_dev, grads_dev, tol=1e-4):
    """pre_dev [R * P] logits and {name: gradient} of the device step.  Returns a report: the worst |err| / (tol-free) scale per
    tensor under both criteria -- `termsum`: max |err| / T (must stay <= tol), `maxnorm`: max |err| / max |ref| -- and the logits'
    worst error relative to max |pre|."""
    R, P = ref['R'], ref['P']
    pre = np.asarray(pre_dev, dtype=np.float64).reshape(R, P)
    e_pre = float(np.abs(pre - ref['pre'][None, :]).max() / max(np.abs(ref['pre']).max(), 1e-300))
    rep = dict(logits_rel_err=e_pre, tensors={}, worst_termsum=0.0, worst_maxnorm=0.0)
    for n, gr in ref['grads'].items():
        got = np.asarray(grads_dev[n], dtype=np.float64).reshape(gr.shape)
        err = np.abs(got - gr)
        # sum |terms| >= |sum terms|: where the sub-pool saw no term at all (a rare input feature, a unit dead on those graphs) the
        # element is held to 1e-4 of its own value; elements that are zero in the reference must be zero to 1e-12 of the tensor
        floor = 1e-12 * max(float(np.abs(gr).max()), 1e-300)
        ts = float((err / np.maximum(np.maximum(ref['T'][n], np.abs(gr)), floor)).max())
        mn = float(err.max() / max(float(np.abs(gr).max()), 1e-300))
        ratio = err / np.maximum(np.maximum(ref['T'][n], np.abs(gr)), floor)
        wi = int(np.argmax(ratio))
        rep['tensors'][n] = dict(termsum=ts, maxnorm=mn, worst_index=[int(v) for v in np.unravel_index(wi, gr.shape)],
                                 worst_got=float(got.flat[wi]), worst_ref=float(gr.flat[wi]), worst_T=float(ref['T'][n].flat[wi]),
                                 tensor_max_abs=float(np.abs(gr).max()))
        rep['worst_termsum'] = max(rep['worst_termsum'], ts)
        rep['worst_maxnorm'] = max(rep['worst_maxnorm'], mn)
    rep['ok'] = bool(rep['worst_termsum'] <= tol and e_pre <= tol)
    return rep
