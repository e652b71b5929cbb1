"""Drop-in modules for /root/reference/libs/spect_conv.py on MI355X.

Same class names, constructor signatures, forward signatures, attributes, ``state_dict`` keys and
parameter initialisation as the reference (SpectConv :23-103, SpectConCatConv :105-165,
ML3Layer :182-212), so the GNNML1/GNNML3 model classes of the experiment scripts build on them
unchanged and a reference checkpoint loads.  The arithmetic runs in the hand-written gfx950 kernels
of libgml_hip.so; inputs must live on the GPU -- there is no CPU fallback.

Differences a caller can observe: none in values beyond fp32 round-off (summation inside a target
row keeps the reference's edge order; the projection is accumulated in a different order);
``edge_index`` may also be a prebuilt ``GraphCSR``; the PyG ``MessagePassing`` base class is not
required (only ``aggr='add'``, ``flow='source_to_target'``, ``node_dim=0`` exist in the reference).
"""
import math
import os

import weakref

import torch
from torch.nn import Parameter

from . import functional as Fn
from .functional import ML3LayerFunction, SpectConvFunction
from .graph import csr_for, GraphCSR, _require_cuda


def glorot(tensor):
    """U(-a, a), a = sqrt(6 / (size(-2) + size(-1)))   (libs/spect_conv.py:13-16)."""
    if tensor is not None:
        stdv = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        tensor.data.uniform_(-stdv, stdv)


def zeros(tensor):
    if tensor is not None:
        tensor.data.fill_(0)


def _check_mp_kwargs(kwargs):
    aggr = kwargs.pop('aggr', 'add')
    flow = kwargs.pop('flow', 'source_to_target')
    node_dim = kwargs.pop('node_dim', 0)
    if aggr != 'add' or flow != 'source_to_target' or node_dim != 0:
        raise NotImplementedError("only aggr='add', flow='source_to_target', node_dim=0 are supported "
                                  "(the reference uses no other setting)")
    if kwargs:
        raise TypeError('unexpected keyword arguments: %s' % sorted(kwargs))
    return aggr, flow, node_dim


def _sorted_values(csr, edge_index, edge_attr, ncols=None):
    """supports in target-sorted order, differentiable w.r.t. edge_attr.  ncols: the layer reads only the first ncols
    columns (the reference indexes edge_attr[:, i] for i < K, libs/spect_conv.py:77, so a wider input is legal); the
    narrowing happens here, outside the autograd Function, so the gradient keeps the caller's shape."""
    _require_cuda(edge_attr, 'edge_attr')
    if edge_attr.dim() == 1:
        edge_attr = edge_attr.view(-1, 1)
    if ncols is not None and edge_attr.size(1) > ncols:
        edge_attr = edge_attr[:, :ncols]
    if edge_attr.size(0) != csr.E:
        raise ValueError('edge_attr has %d rows, edge_index has %d edges' % (edge_attr.size(0), csr.E))
    if edge_attr.dtype != torch.float32:
        raise TypeError('edge_attr must be float32, got %s' % edge_attr.dtype)
    if edge_attr.requires_grad:
        return _SortValues.apply(edge_attr, csr)
    return csr.sort_values(edge_attr.detach())


class _SortValues(torch.autograd.Function):
    @staticmethod
    def forward(ctx, edge_attr, csr):
        ctx.csr = csr
        return csr.sort_values(edge_attr.detach(), cache=False)

    @staticmethod
    def backward(ctx, g):
        return ctx.csr.unsort_values(g.contiguous()), None


class SpectConv(torch.nn.Module):
    r"""out = sum_i (A_i^T x) W_i (+ x W_last if selfconn) + bias, A_i[src, dst] = edge_attr[e, i].

    depthwise: out = ( DS_last*x [selfconn] + (1+DS_0)*H_0 + sum_{i>=1} DS_i*H_i ) W_0 + bias.
    """

    def __init__(self, in_channels, out_channels, K=1, selfconn=True, depthwise=False, bias=True, **kwargs):
        super(SpectConv, self).__init__()
        self.aggr, self.flow, self.node_dim = _check_mp_kwargs(kwargs)
        assert K > 0
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.depthwise = depthwise
        self.selfconn = selfconn
        if self.selfconn:
            K = K + 1
        if self.depthwise:
            self.DSweight = Parameter(torch.Tensor(K, in_channels))
            self.nsup = K
            K = 1
        self.weight = Parameter(torch.Tensor(K, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        zeros(self.bias)
        if self.depthwise:
            zeros(self.DSweight)

    def _effective(self, weight=None, dsw=None):
        """(support weights [S, Fin, Fout], self weight [Fin, Fout] or None) of the equivalent plain form."""
        weight = self.weight if weight is None else weight
        if not self.depthwise:
            if self.selfconn:
                return weight[:-1], weight[-1]
            return weight, None
        ds = self.DSweight if dsw is None else dsw
        nsup = self.nsup - 1 if self.selfconn else self.nsup
        scale = torch.cat([1 + ds[0:1], ds[1:nsup]], 0)                 # [S, Fin]
        w = scale.unsqueeze(-1) * weight[0].unsqueeze(0)                # diag(scale_s) W_0
        wself = ds[-1].unsqueeze(-1) * weight[0] if self.selfconn else None
        return w, wself

    def _mapped(self, x, csr, edge_index, edge_attr, weight=None, bias=None, dsw=None):
        """the layer through the default-branch kernels and (for depthwise) effective weights diag(DS_s) W_0; differentiable"""
        bias = self.bias if (bias is None and weight is None) else bias
        w, wself = self._effective(weight, dsw)
        val = _sorted_values(csr, edge_index, edge_attr, w.size(0))
        out = SpectConvFunction.apply(x, val, w.contiguous(), None if wself is not None else bias, csr, False)
        if wself is not None:
            out = torch.addmm(bias, x, wself) + out if bias is not None else torch.mm(x, wself) + out
        return out

    def _epilogue(self, x, csr, val):
        """depthwise branch in ONE launch on the ring kernel's epilogue (scale-then-one-projection, libs/spect_conv.py:81-91);
        None when the shape is outside that kernel"""
        ds = self.DSweight
        nsup = self.nsup - 1 if self.selfconn else self.nsup
        scale = torch.cat([1 + ds[0:1], ds[1:nsup]] + ([ds[-1:]] if self.selfconn else []), 0).detach().contiguous()
        return Fn.conv_epilogue(csr, x, val, self.weight[0].detach(), self.bias, nsup, 2, scale, self.selfconn, self.out_channels)

    def forward(self, x, edge_index, edge_attr, edge_weight=None, batch=None, lambda_max=None):
        _require_cuda(x, 'x')
        csr = csr_for(edge_index, x.size(0))
        if self.depthwise and Fn.conv_epilogue_applies(self.nsup - 1 if self.selfconn else self.nsup, self.in_channels, self.out_channels):
            return _EpilogueFn.apply(self, csr, edge_index, x, edge_attr, self.weight, self.bias, self.DSweight)
        return self._mapped(x, csr, edge_index, edge_attr)

    def __repr__(self):
        return '{}({}, {}, K={})'.format(self.__class__.__name__, self.in_channels, self.out_channels,
                                         self.weight.size(0))


class SpectConCatConv(torch.nn.Module):
    r"""out = cat_i( (A_i^T x) W_i ) (x W_last first if selfconn) + bias[K*Fout]."""

    def __init__(self, in_channels, out_channels, K, selfconn=True, bias=True, **kwargs):
        super(SpectConCatConv, self).__init__()
        self.aggr, self.flow, self.node_dim = _check_mp_kwargs(kwargs)
        assert K > 0
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.selfconn = selfconn
        if self.selfconn:
            K = K + 1
        self.weight = Parameter(torch.Tensor(K, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(K * out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        zeros(self.bias)

    def _mapped(self, x, csr, edge_index, edge_attr, weight=None, bias=None, dsw=None):
        """the layer through the default-branch kernels and block-structured weights (S x the projection flops); differentiable"""
        bias = self.bias if (bias is None and weight is None) else bias
        weight = self.weight if weight is None else weight
        Kp, Fin, Fout = weight.shape
        S = Kp - 1 if self.selfconn else Kp
        val = _sorted_values(csr, edge_index, edge_attr, S)
        # block-structured weights: support i only feeds output columns [i*Fout, (i+1)*Fout)
        off = 1 if self.selfconn else 0
        wbig = x.new_zeros(S, Fin, Kp * Fout)
        for i in range(S):
            wbig[i, :, (i + off) * Fout:(i + off + 1) * Fout] = weight[i]
        out = SpectConvFunction.apply(x, val, wbig, None, csr, False)
        if self.selfconn:
            pad = x.new_zeros(Fin, Kp * Fout)
            pad[:, :Fout] = weight[-1]
            out = out + torch.mm(x, pad)
        if bias is not None:
            out = out + bias
        return out

    def _epilogue(self, x, csr, val):
        """per-support column blocks written by ONE launch of the ring kernel (projection flops 1 x, libs/spect_conv.py:137-158);
        the selfconn block x W_last is one GEMM into block 0; None when the shape is outside that kernel"""
        Kp, Fin, Fout = self.weight.shape
        S = Kp - 1 if self.selfconn else Kp
        out = Fn.conv_epilogue(csr, x, val, self.weight[:S].detach(), self.bias, S, 1, None, self.selfconn, Kp * Fout)
        if out is not None and self.selfconn:
            blk = torch.mm(x, self.weight[-1].detach())
            out[:, :Fout] = blk + self.bias[:Fout].detach() if self.bias is not None else blk
        return out

    def forward(self, x, edge_index, edge_attr, edge_weight=None, batch=None, lambda_max=None):
        _require_cuda(x, 'x')
        csr = csr_for(edge_index, x.size(0))
        Kp, Fin, Fout = self.weight.shape
        if Fn.conv_epilogue_applies(Kp - 1 if self.selfconn else Kp, Fin, Fout):
            return _EpilogueFn.apply(self, csr, edge_index, x, edge_attr, self.weight, self.bias, None)
        return self._mapped(x, csr, edge_index, edge_attr)

    def __repr__(self):
        return '{}({}, {}, K={})'.format(self.__class__.__name__, self.in_channels, self.out_channels,
                                         self.weight.size(0))


class _EpilogueFn(torch.autograd.Function):
    """SpectConCatConv / depthwise SpectConv: forward = one launch of the ring kernel with the layer's own epilogue
    (gml_spectconv_fwd_epi); backward = autograd of the equivalent mapping onto the default-branch kernels (module._mapped: the
    form round 1-2 also ran the forward on), re-evaluated inside backward.  Falls back to the mapping altogether when the
    library reports the shape unsupported."""

    @staticmethod
    def forward(ctx, module, csr, edge_index, x, edge_attr, weight, bias, dsw):
        S = (weight.size(0) - 1 if module.selfconn else weight.size(0)) if not getattr(module, 'depthwise', False) else \
            (module.nsup - 1 if module.selfconn else module.nsup)
        with torch.no_grad():
            val = _sorted_values(csr, edge_index, edge_attr.detach(), S)
            out = module._epilogue(Fn.rows4(x.detach().contiguous()), csr, val)
            if out is None:                                     # shape outside the epilogue kernel: the mapping
                Fn._path('conv_fwd', 'weight-transform mapping (epilogue kernel does not cover the shape)')
                out = module._mapped(x.detach(), csr, edge_index, edge_attr.detach(), weight.detach(),
                                     None if bias is None else bias.detach(), None if dsw is None else dsw.detach())
        ctx.module, ctx.csr, ctx.edge_index = module, csr, edge_index
        ctx.save_for_backward(x, edge_attr, weight, bias, dsw)
        return out

    @staticmethod
    def backward(ctx, g):
        x, edge_attr, weight, bias, dsw = ctx.saved_tensors
        need = ctx.needs_input_grad[3:]
        ins = [t.detach().requires_grad_(bool(n)) if t is not None else None for t, n in zip((x, edge_attr, weight, bias, dsw), need)]
        with torch.enable_grad():
            out = ctx.module._mapped(ins[0], ctx.csr, ctx.edge_index, ins[1], ins[2], ins[3], ins[4])
        wanted = [t for t in ins if t is not None and t.requires_grad]
        got = iter(torch.autograd.grad(out, wanted, g, allow_unused=True)) if wanted else iter(())
        grads = [next(got) if (t is not None and t.requires_grad) else None for t in ins]
        return (None, None, None) + tuple(grads)


# per ML3Layer instance: (weakref to the output tensor of its latest forward, that forward's functional.ChainToken)
_CHAIN_STATE = weakref.WeakKeyDictionary()
# per ML3Layer instance: (key of supports + weight versions, edge-branch output the head of its stack computed for it)
_EDGE_STASH = weakref.WeakKeyDictionary()


class ML3Layer(torch.nn.Module):
    """One GNNML3 layer: optional per-edge MLP on the supports, relu(SpectConv) || tanh(fc11 x)*tanh(fc12 x)."""

    def __init__(self, learnedge, nedgeinput, nedgeoutput, ninp, nout1, nout2):
        super(ML3Layer, self).__init__()
        self.learnedge = learnedge
        self.nout2 = nout2
        if self.learnedge:
            self.fc1_1 = torch.nn.Linear(nedgeinput, 2 * nedgeinput, bias=False)
            self.fc1_2 = torch.nn.Linear(nedgeinput, 2 * nedgeinput, bias=False)
            self.fc1_3 = torch.nn.Linear(nedgeinput, 2 * nedgeinput, bias=False)
            self.fc1_4 = torch.nn.Linear(4 * nedgeinput, nedgeoutput, bias=False)
        else:
            nedgeoutput = nedgeinput
        self.conv1 = SpectConv(ninp, nout1, nedgeoutput, selfconn=False)
        if nout2 > 0:
            self.fc11 = torch.nn.Linear(ninp, nout2)
            self.fc12 = torch.nn.Linear(ninp, nout2)
        self._chain_prev = ()        # (the ML3Layer whose output tensor IS this layer's x,) -- see chain_after
        self._chain_next = ()        # (the ML3Layer stacked on this one,)

    def chain_after(self, prev):
        """Declare that this layer's input x is `prev`'s output tensor itself and that NOTHING else consumes that tensor
        (Zinc12k.py:338-341: x = conv2(conv1(x, ...), ...)).  The backward then applies prev's relu where this layer produces
        dL/dx and prev's output-stage backward skips its saved output (functional.ChainToken).  Checked per call: only taken
        when x is that very tensor object; with a skip connection or a second reader of prev's output, do not declare it."""
        old = self._chain_prev[0] if self._chain_prev else None
        if old is not None and old._chain_next and old._chain_next[0] is self:
            old._chain_next = ()
        self._chain_prev = (prev,) if prev is not None else ()      # (a tuple: not registered as a submodule)
        if prev is not None:
            prev._chain_next = (self,)
        return self

    def _edge_key(self, val, csr):
        w = (self.fc1_1.weight, self.fc1_2.weight, self.fc1_3.weight, self.fc1_4.weight)
        # val._version: supports rewritten IN PLACE in the same buffer (a static batch buffer) must not be served the branch
        # output of the old ones
        return (val.data_ptr(), val._version, tuple(val.shape), id(csr)) + tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in w)

    def _edge_stack(self, val, csr):
        """(ea_pre, stack) for ML3LayerFunction: this layer's edge-branch output if the head of its stack already computed it
        for exactly these supports and these weight versions; otherwise, for a layer with layers stacked on it that read the
        same supports (chain_after), the request to compute theirs in the same pass."""
        hit = _EDGE_STASH.pop(self, None)
        if hit is not None and hit[0] == self._edge_key(val, csr):
            return hit[1], None
        if not Fn.EDGE_STACK:
            return None, None
        succ, cur = [], self
        while cur._chain_next and len(succ) < 3:
            nxt = cur._chain_next[0]
            if not (nxt.learnedge and all(getattr(nxt, n).weight.shape == getattr(self, n).weight.shape and
                                          getattr(nxt, n).weight.device == val.device for n in ('fc1_1', 'fc1_2', 'fc1_3', 'fc1_4'))):
                break
            succ.append(nxt)
            cur = nxt
        if not succ:
            return None, None
        for m in succ:                                        # whatever an interrupted / partial forward left behind
            _EDGE_STASH.pop(m, None)
        return None, ([tuple(getattr(m, n).weight.detach() for n in ('fc1_1', 'fc1_2', 'fc1_3', 'fc1_4')) for m in succ], [], succ)

    def _chain_args(self, x):
        cin = None
        if self._chain_prev and torch.is_grad_enabled() and x.requires_grad:
            ref, token = _CHAIN_STATE.get(self._chain_prev[0], (None, None))
            if ref is not None and ref() is x:
                cin = token
        return cin, Fn.ChainToken(self.conv1.weight.size(2))

    def forward_pooled(self, x, edge_index, edge_attr, ptr, batch, mean=False):
        """global_add_pool / global_mean_pool (mean=True) of forward(...) in one autograd node: [B, nout1 + nout2].  Same
        values as pooling the layer's output (Zinc12k.py:338-343); the pool's gradient then reaches the layer's backward
        un-expanded ([B, C] + the node -> graph map instead of [N, C]).  ptr [B+1] / batch [N]: int32, grouped per graph."""
        return self.forward(x, edge_index, edge_attr, _pool=(ptr, batch, int(mean)))   # (int: the flag word of gml_segment_sum)

    def forward(self, x, edge_index, edge_attr, _pool=None):
        _require_cuda(x, 'x')
        csr = csr_for(edge_index, x.size(0))
        le, n2 = self.learnedge, self.nout2
        if le and max(self.fc1_1.weight.size(1), self.fc1_4.weight.size(0)) > 16:
            # more than 16 supports (no script goes beyond counting.py's 12; the sr25 sweep of SURVEY s8d raises nfreq to 47):
            # the edge branch as four library GEMMs + elementwise under autograd (libs/spect_conv.py:205-207), the node branch on
            # the fused kernels with the learned supports as its (differentiable) values
            _require_cuda(edge_attr, 'edge_attr')
            ea = edge_attr.to(torch.float32)
            wide = max(self.fc1_1.weight.size(1), self.fc1_4.weight.size(0)) <= 48 and ea.size(1) == self.fc1_1.weight.size(1) \
                and not os.environ.get('GML_EDGE_WIDE_LIB')
            Fn._path('edge', 'one-launch kernel for 17..48 supports' if wide else 'library GEMMs (more than 48 supports)',
                     self.fc1_1.weight.size(1), '-', self.fc1_4.weight.size(0))
            if wide:
                # round 5: gml_edge_mlp_wide_fwd (csrc/gml_edge_wide.hip) -- one launch, weights in LDS, nothing but e and out in HBM
                ev = Fn.EdgeBranchWide.apply(ea.contiguous(), self.fc1_1.weight, self.fc1_2.weight, self.fc1_3.weight, self.fc1_4.weight)
            else:
                # fc1_4 applied to the two halves of its input separately: the concatenation of libs/spect_conv.py:205 is never
                # materialised (at S = 48 it is a 2.5e9-element tensor: torch.cat alone took 83 ms of the layer's 90 on 13 M edges)
                ev = Fn._edge_branch_torch(ea, self.fc1_1.weight, self.fc1_2.weight, self.fc1_3.weight, self.fc1_4.weight)
            val = _sorted_values(csr, edge_index, ev, self.conv1.weight.size(0))
            return ML3LayerFunction.apply(x, val, None, None, None, None, self.conv1.weight, self.conv1.bias,
                                          self.fc11.weight if n2 > 0 else None, self.fc11.bias if n2 > 0 else None,
                                          self.fc12.weight if n2 > 0 else None, self.fc12.bias if n2 > 0 else None,
                                          csr, False, n2, False, *(_pool if _pool is not None else (None, None, False)))   # (no hand-over on this road)
        # learnedge: fc1_1..3 are Linear(nedgeinput, .) -- the width must match, as in the reference; otherwise conv1
        # reads the first K columns only
        # Training with the edge branch: the branch runs in SOURCE order (functional.ML3LayerFunction).  SpectralDesign emits
        # source-sorted edges, so the raw supports in that order are the input tensor itself: no value sort at all.
        raw_src = (le and csr.src_sorted and torch.is_grad_enabled() and edge_attr.dim() == 2 and not edge_attr.requires_grad
                   and edge_attr.dtype == torch.float32 and edge_attr.is_contiguous() and edge_attr.size(0) == csr.E
                   and edge_attr.size(1) == self.fc1_1.weight.size(1) and self.fc1_4.weight.size(0) == edge_attr.size(1)
                   and Fn.ml3_edge_in_source_order(csr, edge_attr.size(1), self.conv1.weight.size(1), self.conv1.weight.size(2)))
        if raw_src:
            _require_cuda(edge_attr, 'edge_attr')
            val = edge_attr.detach()
        else:
            val = _sorted_values(csr, edge_index, edge_attr, None if le else self.conv1.weight.size(0))
        cin, cout = self._chain_args(x)
        ea_pre, stack = self._edge_stack(val, csr) if raw_src else (None, None)
        out = ML3LayerFunction.apply(
            x, val,
            self.fc1_1.weight if le else None, self.fc1_2.weight if le else None,
            self.fc1_3.weight if le else None, self.fc1_4.weight if le else None,
            self.conv1.weight, self.conv1.bias,
            self.fc11.weight if n2 > 0 else None, self.fc11.bias if n2 > 0 else None,
            self.fc12.weight if n2 > 0 else None, self.fc12.bias if n2 > 0 else None,
            csr, le, n2, bool(raw_src), *(_pool if _pool is not None else (None, None, False)), cin, cout, ea_pre,
            stack[:2] if stack is not None else None)
        if stack is not None and len(stack[1]) == len(stack[2]):
            # the layers stacked on this one find their edge-branch outputs here (same supports, same weight versions: checked)
            for m, t in zip(stack[2], stack[1]):
                _EDGE_STASH[m] = (m._edge_key(val, csr), t)
        if _pool is None and torch.is_grad_enabled():
            _CHAIN_STATE[self] = (weakref.ref(out), cout)       # the output tensor of this forward and its hand-over token
        else:
            _CHAIN_STATE.pop(self, None)
        return out


__all__ = ['SpectConv', 'SpectConCatConv', 'ML3Layer', 'GraphCSR', 'glorot', 'zeros']
