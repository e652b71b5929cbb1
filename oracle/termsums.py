"""Term sums of ML3Layer (TEST INFRASTRUCTURE, see oracle/__init__.py): for every output and gradient of the layer, the sum of the
absolute values of the terms it is a sum of -- the scale an elementwise round-off criterion |got - ref| <= tol * T is measured on.
Moved here from tests/test_gpu_parity.py (round 5) so that oracle/parity_at_size.py can use it layer by layer."""
import torch

from .spect_conv_oracle import propagate_add


def ml3_termsums(meta, x, ea, ei, g, P, Tx=None, Tg=None):
    """fp64 first-order error-bound propagation through ML3Layer (libs/spect_conv.py:204-212) and its backward: for every output
    and gradient the sum of the absolute values of the terms it is a sum of, with the term sums of the intermediate values
    carried along (T of a value that is itself a sum >= |value|); elementwise functions pass T on scaled by |f'| and add their
    own |value|.  meta = (learnedge, ne, neo, ninp, nout1, nout2); x, ea, g (gradient at the layer output) float64 tensors, ei
    int64 [2, E], P {parameter name: float64 tensor}.  Tx / Tg (optional): the term sums of x and of g themselves when they are
    results of earlier computations (a lower layer's output, a higher layer's input gradient) -- they then stand in for |x| / |g|
    wherever a magnitude enters a sum, which carries the round-off of the rest of the model through this layer (default: x and g are
    exact data, T = |value|).  Returns dict name -> T array (numpy)."""
    learnedge, ne, neo, ninp, nout1, nout2 = [int(v) for v in meta]
    xa, out = (x.abs() if Tx is None else torch.maximum(Tx, x.abs())), {}
    ga_all = g.abs() if Tg is None else torch.maximum(Tg, g.abs())
    if learnedge:
        W1, W2, W3, W4 = P['fc1_1.weight'], P['fc1_2.weight'], P['fc1_3.weight'], P['fc1_4.weight']
        e1, e2, e3 = ea @ W1.t(), ea @ W2.t(), ea @ W3.t()
        Te1, Te2, Te3 = ea.abs() @ W1.abs().t(), ea.abs() @ W2.abs().t(), ea.abs() @ W3.abs().t()
        th2, th3 = torch.tanh(e2), torch.tanh(e3)
        d2, d3 = 1 - th2 ** 2, 1 - th3 ** 2
        m1 = (e1 > 0).double()
        t = torch.cat([torch.relu(e1), th2 * th3], 1)
        Tt = torch.cat([Te1 * m1, Te2 * d2 * th3.abs() + th2.abs() * Te3 * d3 + (th2 * th3).abs()], 1)
        z = t @ W4.t()
        mz = (z > 0).double()
        v, Tv = torch.relu(z), (Tt @ W4.abs().t()) * mz
    else:
        v, Tv = ea, ea.abs()
    Wc = P['conv1.weight']
    S = Wc.size(0)
    H = [propagate_add(x, ei, v[:, s]) for s in range(S)]
    TH = [propagate_add(xa, ei, Tv[:, s]) for s in range(S)]
    cc = sum(H[s] @ Wc[s] for s in range(S)) + P['conv1.bias']
    Tc = sum(TH[s] @ Wc[s].abs() for s in range(S)) + P['conv1.bias'].abs()
    mc = (cc > 0).double()
    Tout = [Tc]
    Ga = ga_all[:, :nout1] * mc
    out['conv1.weight'] = torch.stack([TH[s].t() @ Ga for s in range(S)])
    out['conv1.bias'] = Ga.sum(0)
    back = [Ga @ Wc[s].abs().t() for s in range(S)]                        # [N, Fin] per support
    Tdv = torch.stack([(xa[ei[0]] * back[s][ei[1]]).sum(1) for s in range(S)], 1)
    Tgx = torch.zeros_like(xa)
    for s in range(S):
        Tgx.index_add_(0, ei[0], Tv[:, s:s + 1] * back[s][ei[1]])
    if nout2 > 0:
        W11, b11, W12, b12 = P['fc11.weight'], P['fc11.bias'], P['fc12.weight'], P['fc12.bias']
        n1, n2 = x @ W11.t() + b11, x @ W12.t() + b12
        Tn1, Tn2 = xa @ W11.abs().t() + b11.abs(), xa @ W12.abs().t() + b12.abs()
        t1, t2 = torch.tanh(n1), torch.tanh(n2)
        q1, q2 = 1 - t1 ** 2, 1 - t2 ** 2
        Tout.append(Tn1 * q1 * t2.abs() + t1.abs() * Tn2 * q2 + (t1 * t2).abs())
        gb = ga_all[:, nout1:]
        # (tanh' = 1 - tanh^2 is itself a difference: ITS term sum is 1 + tanh^2, not its value -- a saturated unit's derivative
        #  carries the absolute round-off of the 1, in the reference's fp32 tanh_backward as much as here)
        Tg1 = gb * (t2.abs() * (1 + t1 ** 2) + Tn2 * q2 * q1 + t2.abs() * 2 * t1.abs() * q1 * Tn1)
        Tg2 = gb * (t1.abs() * (1 + t2 ** 2) + Tn1 * q1 * q2 + t1.abs() * 2 * t2.abs() * q2 * Tn2)
        Tgx = Tgx + Tg1 @ W11.abs() + Tg2 @ W12.abs()
        out['fc11.weight'], out['fc11.bias'] = Tg1.t() @ xa, Tg1.sum(0)
        out['fc12.weight'], out['fc12.bias'] = Tg2.t() @ xa, Tg2.sum(0)
    out['out'], out['g_x'] = torch.cat(Tout, 1), Tgx
    if learnedge:
        Tdz = Tdv * mz
        out['fc1_4.weight'] = Tdz.t() @ Tt
        Tdt = Tdz @ W4.abs()
        k = e1.size(1)
        Tde1 = Tdt[:, :k] * m1
        Tdp = Tdt[:, k:]
        Tde2 = Tdp * (th3.abs() * (1 + th2 ** 2) + Te3 * d3 * d2 + th3.abs() * 2 * th2.abs() * d2 * Te2)
        Tde3 = Tdp * (th2.abs() * (1 + th3 ** 2) + Te2 * d2 * d3 + th2.abs() * 2 * th3.abs() * d3 * Te3)
        out['fc1_1.weight'], out['fc1_2.weight'], out['fc1_3.weight'] = Tde1.t() @ ea.abs(), Tde2.t() @ ea.abs(), Tde3.t() @ ea.abs()
        out['g_edge_attr'] = Tde1 @ W1.abs() + Tde2 @ W2.abs() + Tde3 @ W3.abs()
    else:
        out['g_edge_attr'] = Tdv
    return {n: t_.numpy() for n, t_ in out.items()}


