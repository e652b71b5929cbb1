"""Relu margins of an ML3Layer case (TEST INFRASTRUCTURE -- see oracle/__init__.py).

A relu whose argument sits within rounding distance of zero makes the GRADIENT of a layer ambiguous:
two correct evaluations (fp64 here, fp32 in the reference, split-bf16 matrix cores in the HIP path —
all within the 1e-4 output tolerance) can land on different sides and then differ by a whole
gout-sized term in every gradient downstream.  Randomised parity tests therefore need inputs
whose relu arguments all keep a margin.  The three relus of the layer
(/root/reference/libs/spect_conv.py:206-209): the first edge branch relu(fc1_1 e), the edge
output relu(fc1_4 [...]) and the node output relu(conv1(...)).

`margins()` returns, per relu, |argument| / sum|terms| evaluated in fp64 (the sum of absolute
terms bounds the rounding error of any summation order; it is floored at 1e-2 of the largest such
sum in the tensor, because fast tanh formulas are accurate to an ABSOLUTE 1e-7, so an argument
built only from tiny terms is not resolved relative to itself); `make_safe()` re-draws the edge
attribute rows whose edge-branch margins are too small and returns a 0/1 mask for the node
outputs whose margin is too small (the test multiplies its output gradient by it).
"""
import torch
import torch.nn.functional as F

from .spect_conv_oracle import spectconv_forward


def _ratio(arg, scale):
    """|arg| / max(scale, 1e-2 max(scale)); 1 where every term is exactly zero (both sides compute an exact 0)."""
    floor = 1e-2 * float(scale.max()) if scale.numel() else 0.0
    return torch.where(scale > 0, arg.abs() / scale.clamp_min(max(floor, 1e-300)), torch.ones_like(scale))


def _edge_parts(ea, p):
    w1, w2, w3, w4 = (p['fc1_%d.weight' % i].double() for i in (1, 2, 3, 4))
    h1 = F.linear(ea, w1)
    s1 = F.linear(ea.abs(), w1.abs())
    h = torch.cat([F.relu(h1), torch.tanh(F.linear(ea, w2)) * torch.tanh(F.linear(ea, w3))], 1)
    o = F.linear(h, w4)
    so = F.linear(h.abs(), w4.abs())
    return h1, s1, o, so


def margins(x, edge_index, edge_attr, p, learnedge):
    """p: state_dict-style parameter dict.  Returns (m_edge [E] or None, m_node [N, nout1]); entries whose terms
    are all exactly zero (isolated rows, all-zero supports) report margin 1: both sides compute an exact 0."""
    x, ea = x.double(), edge_attr.double()
    m_edge = None
    if learnedge:
        h1, s1, o, so = _edge_parts(ea, p)
        r1, ro = _ratio(h1, s1), _ratio(o, so)
        m_edge = torch.minimum(r1.min(1).values, ro.min(1).values) if ea.size(0) else r1.new_zeros(0)
        ea = F.relu(o)
    cw = p['conv1.weight'].double()
    cb = p['conv1.bias'].double() if p.get('conv1.bias') is not None else None
    pre = spectconv_forward(x, edge_index, ea, cw, cb)
    sc = spectconv_forward(x.abs(), edge_index, ea.abs(), cw.abs(), cb.abs() if cb is not None else None)
    return m_edge, _ratio(pre, sc)


def make_safe_edges(edge_attr, w1, w2, w3, w4, margin=2e-4, scale=1.0, max_rounds=20, generator=None):
    """edge_attr with the rows re-drawn whose two edge-branch relu arguments come within `margin` of zero."""
    p = {'fc1_1.weight': w1, 'fc1_2.weight': w2, 'fc1_3.weight': w3, 'fc1_4.weight': w4}
    ea = edge_attr.clone()
    for _ in range(max_rounds):
        h1, s1, o, so = _edge_parts(ea.double(), p)
        r1, ro = _ratio(h1, s1), _ratio(o, so)
        bad = (torch.minimum(r1.min(1).values, ro.min(1).values) < margin).nonzero().flatten()
        if bad.numel() == 0:
            return ea
        ea[bad] = (torch.randn(bad.numel(), ea.size(1), generator=generator) * scale).to(ea.dtype)
    raise RuntimeError('could not find relu-safe edge attributes')


def make_safe(x, edge_index, edge_attr, p, learnedge, margin=2e-4, scale=0.5, max_rounds=20, generator=None):
    """Returns (edge_attr', node_mask): edge_attr with the offending rows re-drawn (N(0, scale^2), like the callers
    draw them) until every edge-branch relu keeps `margin`, and a float mask [N, nout1] that is 0 on the node
    outputs that do not."""
    ea = edge_attr.clone()
    for _ in range(max_rounds):
        m_edge, m_node = margins(x, edge_index, ea, p, learnedge)
        if m_edge is None:
            break
        bad = (m_edge < margin).nonzero().flatten()
        if bad.numel() == 0:
            break
        ea[bad] = (torch.randn(bad.numel(), ea.size(1), generator=generator) * scale).to(ea.dtype)
    else:
        raise RuntimeError('could not find relu-safe edge attributes')
    return ea, (m_node >= margin).to(x.dtype)
