"""Dense-block evaluation of SpectConv for batches of equal-size graphs with near-dense masks (SURVEY s8f rank 3:
MNIST-75, recfield >= 3 -- the 4-hop mask of a 75-node superpixel graph is ~80 % full, so block-CSR degenerates to dense
blocks; the TF reference formulates the layer exactly this way, libs/layers_tf.py:231-236).

    Hcat[b n + j, s Fin + f] = sum_i D[b, s][j, i] X[b n + i, f]        gml_dense_support_mm  (csrc/gml_dense.hip)
    out                      = Hcat Wcat + bias                         one tall GEMM  [B n, S Fin] x [S Fin, Fout]

The batched support product is the hand-written part (one workgroup per graph, the supports streamed once from bf16
(hi, lo) images, bf16x3 on the matrix cores); its adjoint d X = sum_s D_s^T d Hcat_s is the same kernel on the transposed
images.  The tall GEMM and its two gradient GEMMs contract over S Fin = 384 .. 768 or over all nodes of the batch -- plain
library GEMM shapes, left to rocBLAS through torch.  ``GML_DENSE_LIB=1`` (or the exact-product mode GML_F32_MFMA=1, which
this kernel does not have) evaluates the support product with torch.bmm instead: the round-1 path, kept as the A/B
baseline of tools/bench_mnist.py.  Same parameters and values as ``SpectConv`` on the sparse path
(tests/test_gpu_parity.py compares both with the oracle and with the TF-graph fixture).
"""
import os

import torch

from . import _lib
from . import functional as Fn
from .graph import _ptr, _stream

USE_LIBRARY = os.environ.get('GML_DENSE_LIB', '0') not in ('0', '')


class DenseSupports(object):
    """Per-batch (in practice per-data-set: the supports are constants) dense support blocks of B graphs of n nodes.

    blocks [B, S, n, n] fp32 (row = target node j, column = source node i; only kept for the library path);
    fwd / bwd: bf16 (hi, lo) images [B, S, 2, n, KP] of the blocks / of their transposes (gml_dense_pack)."""

    def __init__(self, blocks, keep_blocks):
        B, S, n, _ = blocks.shape
        self.B, self.S, self.n = int(B), int(S), int(n)
        self.KP = (self.n + 31) // 32 * 32
        self.blocks = blocks if keep_blocks else None
        self.fwd = self.bwd = None
        if not keep_blocks:
            self.fwd, self.bwd = self._pack(blocks, 0), self._pack(blocks, 1)

    def _pack(self, blocks, transpose):
        img = torch.empty(self.B, self.S, 2, self.n, self.KP, dtype=torch.int16, device=blocks.device)
        _lib.call('gml_dense_pack', _ptr(blocks), _ptr(img), self.B * self.S, self.n, self.KP, transpose, _stream(blocks.device))
        return img


def _library():
    return USE_LIBRARY or Fn.exact_mode()


def dense_supports(edge_index2, edge_attr2, ptr, n):
    """DenseSupports of a batch whose graphs all have exactly n nodes (ptr [B+1]): block[b, s][j, i] = value of edge
    i -> j of support s (the mask lists every (i, j) once)."""
    B = int(ptr.numel() - 1)
    S = int(edge_attr2.size(1))
    if B * n != int(ptr[-1]):
        raise ValueError('dense blocks need equal-size graphs: %d graphs, %d nodes, n=%d' % (B, int(ptr[-1]), n))
    if n > 96:
        raise ValueError('dense blocks are built for n <= 96 nodes per graph, got %d' % n)
    src, dst = edge_index2[0], edge_index2[1]
    b = torch.div(src, n, rounding_mode='floor')
    i, j = src - b * n, dst - b * n
    blocks = torch.zeros(B, S, n, n, dtype=torch.float32, device=edge_attr2.device)
    blocks[b, :, j, i] = edge_attr2.float()
    return DenseSupports(blocks, keep_blocks=_library())


def support_mm(img, act, sup, F, sa, so, sum_s):
    """gml_dense_support_mm on packed images: act [B n, >= F (+ s sa)] -> [B n, F] (sum_s) or [B n, S so]."""
    act = act.contiguous()
    out = torch.empty(sup.B * sup.n, F if sum_s else sup.S * so, dtype=torch.float32, device=act.device)
    _lib.call('gml_dense_support_mm', _ptr(img), _ptr(act), int(act.stride(0)), int(sa), _ptr(out), int(out.stride(0)), int(so),
              int(bool(sum_s)), sup.B, sup.S, sup.n, sup.KP, int(F), _stream(act.device))
    return out


class _SupportProduct(torch.autograd.Function):
    """Hcat = [D_0 X | ... | D_{S-1} X] per graph; gradient d X = sum_s D_s^T d Hcat_s (the supports are constants)."""

    @staticmethod
    def forward(ctx, x, sup):
        ctx.sup, ctx.Fin = sup, int(x.size(1))
        Fn._path('dense', 'support product fwd (bf16x3 HIP)', sup.S, ctx.Fin, ctx.Fin)
        return support_mm(sup.fwd, x, sup, ctx.Fin, 0, ctx.Fin, False)

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return None, None
        Fn._path('dense', 'support product bwd (bf16x3 HIP)', ctx.sup.S, ctx.Fin, ctx.Fin)
        return support_mm(ctx.sup.bwd, g, ctx.sup, ctx.Fin, ctx.Fin, 0, True), None


def _splits(rows, lo=1024):
    """number of row slabs of the weight-gradient GEMM: the largest divisor of rows <= 64 that leaves slabs of >= lo rows"""
    for p in range(64, 1, -1):
        if rows % p == 0 and rows // p >= lo:
            return p
    return 1


class _TallGemm(torch.autograd.Function):
    """out = h w + bias for h [B n, S Fin] (hundreds of thousands of rows), w [S Fin, Fout].  Forward and d h are library
    GEMMs as they are; d w = h^T g contracts over ALL rows into a 768 x 128 result, which rocBLAS tiles into ~24
    workgroups (352 us at 76,800 rows, profiles/r02_mnist75_*): it is evaluated as a batched GEMM over row slabs
    (every CU busy) followed by a fixed-order sum of the slab results."""

    @staticmethod
    def forward(ctx, h, w, bias):
        ctx.save_for_backward(h, w)
        ctx.has_bias = bias is not None
        return h.mm(w) if bias is None else torch.addmm(bias, h, w)

    @staticmethod
    def backward(ctx, g):
        h, w = ctx.saved_tensors
        g = g.contiguous()
        dh = g.mm(w.t()) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            rows = int(h.size(0))
            P = _splits(rows)
            dw = torch.bmm(h.view(P, rows // P, h.size(1)).transpose(1, 2), g.view(P, rows // P, g.size(1))).sum(0) if P > 1 else h.t().mm(g)
        db = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dh, dw, db


# Weight gradient: by default Hcat is written in the forward and dW = Hcat^T g runs as a row-slab batched library GEMM (rounds 2-4).
# GML_DENSE_DW_HIP=1: gml_dense_conv_bwd_w (round 5) -- no Hcat, the support product recomputed per graph and contracted with g on the
# matrix cores, hand-written.  Parity-tested, but SLOWER (MNIST-75 step at 4,096 graphs: 5.5 ms against 3.6 ms; 1,024 graphs: 1.55 against
# 1.19): a workgroup per (support, 64-column block, graph slice) recomputes the product per column block and pays the full load
# latency per graph (208 VGPRs at Fin = 128: one workgroup per CU).  What it needs is in DESIGN s8.
DW_LIBRARY = os.environ.get('GML_DENSE_DW_HIP', '0') in ('0', '')
DW_GEMM_LIB = os.environ.get('GML_DENSE_DW_GEMM_LIB', '0') not in ('0', '')   # A/B: the dW GEMM Hcat^T g through the library instead of gml_xty_wide
CHAIN = os.environ.get('GML_DENSE_CHAIN', '1') not in ('0', '')      # projection chained behind the support product (gml_dense_conv_fwd)
CHAIN_BWD = os.environ.get('GML_DENSE_CHAIN_BWD', '1') not in ('0', '')   # and the dX path: projection in front of the transposed product


class _DenseConv(torch.autograd.Function):
    """out = sum_s (D_s X) W_s + bias in ONE launch: the support product's accumulators are the projection's operand
    (csrc/gml_dense.hip: gml_k_dense_conv_fwd), as libs/layers_tf.py:231-236 forms the layer.  Hcat is written only as the
    saved tensor of the weight gradient dW = Hcat^T g (row-slab batched GEMM, see _TallGemm); dX = sum_s D_s^T (g W_s^T) is
    the library GEMM g Wcat^T followed by the support product on the transposed images."""

    @staticmethod
    def forward(ctx, x, weight, bias, sup, relu=False):
        S, Fin, Fout = weight.shape
        x = x.contiguous()
        dev = x.device
        rows = sup.B * sup.n
        wimg = torch.empty(int(_lib.lib().gml_dense_wimg_elems(S, Fin, Fout)), dtype=torch.int16, device=dev)
        _lib.call('gml_dense_pack_w', _ptr(weight.contiguous()), _ptr(wimg), S, Fin, Fout, _stream(dev))
        need_h = ctx.needs_input_grad[1] and DW_LIBRARY           # (round 5: the weight gradient recomputes the support product -- no Hcat)
        hcat = torch.empty(rows, S * Fin, dtype=torch.float32, device=dev) if need_h else None
        out = torch.empty(rows, Fout, dtype=torch.float32, device=dev)
        Fn._path('dense', 'support product + projection chained (bf16x3 HIP)', S, Fin, Fout)
        # algorithmic bytes: the packed support images (bf16 hi + lo), x, out, and Hcat when the weight gradient wants it
        q_img = sup.B * S * 2 * sup.n * sup.KP * 2
        with Fn._Timed('dense_conv_fwd', q_img + 4 * rows * (Fin + Fout + (S * Fin if need_h else 0)), 2 * sup.B * S * sup.n * sup.n * Fin + 2 * rows * S * Fin * Fout,
                       q2=q_img + 4 * rows * (Fin + Fout)):                 # compulsory: without the Hcat this design writes for its own dW
            _lib.call('gml_dense_conv_fwd', _ptr(sup.fwd), _ptr(x), int(x.stride(0)), _ptr(wimg), _ptr(bias), _ptr(out), Fout,
                      _ptr(hcat), sup.B, S, sup.n, sup.KP, Fin, Fout, 1 if relu else 0, _stream(dev))
        # relu (round 5): applied in the kernel's epilogue (libs/layers_tf.py:238: act(output)); the backward's mask and the bias
        # gradient come out of ONE pass over g and the saved output (functional.ml3_split_bwd) -- before: a relu launch forward, a
        # threshold launch and a column sum backward, 1.6 GB of elementwise traffic per MNIST-75 step at 4,096 graphs
        ctx.save_for_backward(hcat if hcat is not None else x, weight, out if relu else None)
        ctx.sup, ctx.has_bias, ctx.has_h, ctx.relu = sup, bias is not None, hcat is not None, bool(relu)
        return out

    @staticmethod
    def backward(ctx, g):
        hcat, weight, yout = ctx.saved_tensors
        sup = ctx.sup
        S, Fin, Fout = weight.shape
        g = g.contiguous()
        dx = dw = db = None
        if ctx.relu:
            r = Fn.ml3_split_bwd(g, yout, Fout, need_dcb=ctx.has_bias and ctx.needs_input_grad[2])
            if r is not None:
                g = r[0]
                db = r[2]
                if not g.is_contiguous():
                    g = g.contiguous()
            else:
                g = g * (yout > 0)
        if ctx.needs_input_grad[0]:
            if CHAIN_BWD:
                # dX = sum_s D_s^T (g W_s^T) in one launch: d Hcat stays on the CU (gml_k_dense_conv_bwdx)
                dev = g.device
                wimgT = torch.empty(int(_lib.lib().gml_dense_wimgt_elems(S, Fin, Fout)), dtype=torch.int16, device=dev)
                _lib.call('gml_dense_pack_wt', _ptr(weight.contiguous()), _ptr(wimgT), S, Fin, Fout, _stream(dev))
                dx = torch.empty(sup.B * sup.n, Fin, dtype=torch.float32, device=dev)
                Fn._path('dense', 'projection + support product chained, backward (bf16x3 HIP)', S, Fin, Fout)
                with Fn._Timed('dense_conv_bwd_x', sup.B * S * 2 * sup.n * sup.KP * 2 + 4 * sup.B * sup.n * (Fin + Fout),
                               2 * sup.B * S * sup.n * sup.n * Fin + 2 * sup.B * sup.n * S * Fin * Fout):
                    _lib.call('gml_dense_conv_bwd_x', _ptr(sup.bwd), _ptr(g), int(g.stride(0)), _ptr(wimgT), _ptr(dx), Fin,
                              sup.B, S, sup.n, sup.KP, Fin, Fout, _stream(dev))
            else:
                dh = g.mm(weight.reshape(S * Fin, Fout).t())
                dx = support_mm(sup.bwd, dh, sup, Fin, Fin, 0, True)
        if ctx.needs_input_grad[1] and not ctx.has_h:
            # dW = sum over the graphs of (D_s X)^T g: the support product recomputed on the matrix cores, contracted with g over the graph's
            # rows, no Hcat in HBM and no library GEMM (gml_dense_conv_bwd_w, csrc/gml_dense.hip)
            x, dev = hcat, g.device
            Fn._path('dense', 'weight gradient: support product recomputed + row contraction (bf16x3 HIP)', S, Fin, Fout)
            ws = torch.empty(max(int(_lib.lib().gml_dense_dw_workspace_bytes(sup.B, S, Fin, Fout)), 4), dtype=torch.uint8, device=dev)
            dw = torch.empty(S, Fin, Fout, dtype=torch.float32, device=dev)
            _lib.call('gml_dense_conv_bwd_w', _ptr(sup.fwd), _ptr(x), int(x.stride(0)), _ptr(g), int(g.stride(0)), _ptr(dw), sup.B, S, sup.n,
                      sup.KP, Fin, Fout, _ptr(ws), ws.numel(), _stream(dev))
        elif ctx.needs_input_grad[1]:
            rows = int(hcat.size(0))
            P = _splits(rows)
            dw = None
            if not DW_GEMM_LIB:                                   # round 5: Hcat^T g on the bf16 matrix cores (gml_xty_wide), not a library GEMM
                with Fn._Timed('dense_dw_xty_wide', 4 * rows * (S * Fin + Fout), 2 * rows * S * Fin * Fout, q2=4 * rows * (Fin + Fout)):
                    dw = Fn.xty_wide(hcat, g)
                dw = dw.view(S, Fin, Fout) if dw is not None else None
            if dw is None:
                with Fn._Timed('dense_dw_library_gemm', 4 * rows * (S * Fin + Fout), 2 * rows * S * Fin * Fout):
                    dw = (torch.bmm(hcat.view(P, rows // P, S * Fin).transpose(1, 2), g.view(P, rows // P, Fout)).sum(0) if P > 1
                          else hcat.t().mm(g)).view(S, Fin, Fout)
        if ctx.has_bias and ctx.needs_input_grad[2] and db is None:
            db = g.sum(0)
        return dx, dw, db, None, None


def spectconv_dense(x, sup, weight, bias, n, relu=False):
    """x [B*n, Fin], sup from dense_supports, weight [S, Fin, Fout] -> [B*n, Fout] = act(sum_s (D_s x) W_s + bias), act = relu or none."""
    S, Fin, Fout = weight.shape
    if sup.n != n or sup.S != S or x.size(0) != sup.B * n:
        raise ValueError('supports [%d graphs, S=%d, n=%d] do not match x %s / weight %s' %
                         (sup.B, sup.S, sup.n, tuple(x.shape), tuple(weight.shape)))
    if sup.blocks is not None:                         # library path (GML_DENSE_LIB=1 / GML_F32_MFMA=1)
        Fn._path('dense', 'support product (torch.bmm fp32)', S, Fin, Fout)
        h = torch.bmm(sup.blocks.view(sup.B, S * n, n), x.view(sup.B, n, Fin))
        h = h.view(sup.B, S, n, Fin).permute(0, 2, 1, 3).reshape(sup.B * n, S * Fin)
    else:
        if Fin > 128:
            raise ValueError('the dense-block kernel covers Fin <= 128, got %d' % Fin)
        if CHAIN and Fout <= 128:
            return _DenseConv.apply(x, weight, bias, sup, relu)
        h = _SupportProduct.apply(x, sup)
    out = _TallGemm.apply(h, weight.reshape(S * Fin, Fout), bias)
    return torch.relu(out) if relu else out
