"""Autograd functions over the C ABI of libgml_hip.so.

Forward of one SpectConv (libs/spect_conv.py:64-96) is ONE fused launch (gml_spectconv_fwd / gml_ml3_fwd); backward is ONE
fused launch organised by source rows (gml_spectconv_bwd*: dX = sum_s A_s (G W_s^T), dval[e, s] = <X[src] W_s, G[dst]>,
dW_s = X^T (A_s G), no atomics) plus the one-pass output stage (gml_ml3_split_bwd*: relu mask, bias sums, Hadamard branch).
Only shapes outside every compiled kernel class take the unfused composition (_conv_backward: transposed forward + SpMM +
SDDMM + two GEMMs; GML_VERBOSE=1 shows which road a call took).
Every tensor handed to the library is fp32, contiguous and on the current CUDA(HIP) device; the
launches go to torch's current stream, so they order with the surrounding torch ops and are
captured by torch.cuda.graphs like any other kernel.
"""
import ctypes

import torch

from . import _lib
from .graph import _ptr, _stream, _require_cuda


# ---------------------------------------------------------------------------- live kernel timing
# bench.py sets PROFILE = {} for the timed region: every tagged launch is then bracketed by two HIP
# events recorded on the stream the kernel runs on, with its algorithmic bytes / flops.
PROFILE = None

# Projection arithmetic of the fused kernels: default = bf16x3 split on the bf16 matrix cores (fp32-class,
# ~1e-6 of the output scale); F32_MFMA = True (or env GML_F32_MFMA=1) = f32-input MFMA, bit-identical to fmaf.
import os as _os
import collections as _collections
import weakref as _weakref
_linear = torch.nn.functional.linear
F32_MFMA = _os.environ.get('GML_F32_MFMA', '0') not in ('0', '')


import threading as _threading
_TLS = _threading.local()


# Diagnostic switches (tools/precision_diag.py): EXACT_ONLY = None or a set of kernel families ('edge', 'conv') -- exact products are then
# taken only by those families; FORCE_BWD_EXACT = None / True / False overrides the arithmetic a backward inherits from its forward.
EXACT_ONLY = None
FORCE_BWD_EXACT = None


def _bwd_exact(ctx):
    return ctx.exact if FORCE_BWD_EXACT is None else bool(FORCE_BWD_EXACT)


def exact_mode(family='conv'):
    """True when the launches of THIS thread run on exact f32 products: the process-wide default F32_MFMA (environment, tests,
    bench) or an exact_products() scope of this thread.  The scope is thread-local: autograd runs a layer's backward on its own
    worker thread, and a scope entered there (ctx.exact) must not flip the arithmetic of a forward another thread is in the
    middle of."""
    on = bool(F32_MFMA) or getattr(_TLS, 'exact', 0) > 0
    return on and (EXACT_ONLY is None or family in EXACT_ONLY)


class exact_products(object):
    """``with exact_products():`` -- the launches of this thread inside the scope run on exact f32 products (a layer whose gradients
    a BatchNorm backward amplifies: models.GNNML3(bn=True) runs its FIRST layer so; the layer's backward re-enters the scope through
    ctx.exact).  Nestable; per thread."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        if self.on:
            _TLS.exact = getattr(_TLS, 'exact', 0) + 1

    def __exit__(self, *a):
        if self.on:
            _TLS.exact -= 1


_FOLDS = None                      # the open deferred_folds scope's job list (process-wide: autograd runs backward on its own threads)
_FOLDS_LOCK = _threading.Lock()


class deferred_folds(object):
    """``with deferred_folds(model.parameters()): loss.backward()`` -- every weight-gradient kernel launched inside leaves its
    per-workgroup partial sums unfolded (include/gml.h "Deferred folds") and ONE launch folds them all when the scope closes
    (bit-identical sums: the same order).  For the reference's batch size, where a step is a chain of tiny launches (twelve folds at
    ZINC's four layers).  The gradient tensors autograd hands to the parameters are FILLED AT SCOPE EXIT: read .grad (optimizer,
    all-reduce) after the ``with`` block, never inside it.

    REQUIREMENT: every parameter's ``.grad`` is None when the backward runs (``zero_grad(set_to_none=True)``), so that autograd ADOPTS
    the not-yet-filled tensor as ``.grad``.  With an existing ``.grad`` (``zero_grad(set_to_none=False)``, micro-batch accumulation)
    AccumulateGrad would add the unfilled memory into it inside the scope and the fold would never reach it.  Pass the
    parameters the backward reaches: if any of them already holds a gradient the scope is INERT -- the folds run undeferred, launch by
    launch, and accumulation behaves as without the scope.  ``deferred_folds(None)`` = the caller vouches for the requirement.
    The scope is process-wide (the backward runs on autograd's worker threads, not on the thread that opened it): one at a time, one
    device, not nestable."""

    def __init__(self, params):
        self.active = params is None or all(p.grad is None for p in params)

    def __enter__(self):
        global _FOLDS
        if not self.active:
            return self
        with _FOLDS_LOCK:
            assert _FOLDS is None, 'deferred_folds scopes do not nest / overlap'
            _FOLDS = []
        return self

    def __exit__(self, exc_type, *a):
        global _FOLDS
        if not self.active:
            return
        with _FOLDS_LOCK:
            jobs, _FOLDS = _FOLDS, None
        if exc_type is None:
            flush_folds(jobs)


class _FoldQueue(object):
    def __init__(self, jobs):
        self.jobs = jobs

    def append(self, job):
        with _FOLDS_LOCK:
            self.jobs.append(job)


def _fold_queue():
    jobs = _FOLDS
    return _FoldQueue(jobs) if jobs is not None else None


def flush_folds(jobs):
    """jobs: [(partial tensor, nparts, n, [(dst tensor or None, count), ...])] -> gml_fold_many in chunks of 16"""
    if not jobs:
        return
    dev = jobs[0][0].device
    with torch.cuda.device(dev):
        for c0 in range(0, len(jobs), _lib.GML_FOLD_MAX_JOBS):
            chunk = jobs[c0:c0 + _lib.GML_FOLD_MAX_JOBS]
            arr = (_lib.FoldJob * len(chunk))()
            for a, (part, nparts, n, dsts) in zip(arr, chunk):
                a.partial, a.nparts, a.n = part.data_ptr(), int(nparts), int(n)
                for k in range(5):
                    t, c = dsts[k] if k < len(dsts) else (None, 0)
                    a.dst[k] = t.data_ptr() if t is not None else None
                    a.ndst[k] = int(c)
            with _Timed('fold'):
                _lib.call('gml_fold_many', ctypes.addressof(arr), len(chunk), _stream(dev))


EDGE_VALU = _os.environ.get('GML_EDGE_VALU', '0') == '1'


# GML_VERBOSE=1: count which kernel family every layer call takes (shapes off the compiled set degrade in SPEED, never in
# results -- this is how a caller sees it).  PATHS maps e.g. 'conv_bwd: fused bf16x3 (S=8 Fin=32 Fout=30)' -> calls;
# printed at exit, read by tools/bench_configs.py to assert "no unfused fallback on any BASELINE config".
VERBOSE = _os.environ.get('GML_VERBOSE', '0') not in ('0', '')
PATHS = {}


def _path(kind, what, S=None, Fin=None, Fout=None):
    if VERBOSE:
        key = '%s: %s' % (kind, what) + ((' (S=%s Fin=%s Fout=%s)' % (S, Fin, Fout)) if S is not None else '')
        PATHS[key] = PATHS.get(key, 0) + 1


if VERBOSE:
    import atexit as _atexit
    import sys as _sys
    _atexit.register(lambda: [print('[gml path] %5d x %s' % (n, k), file=_sys.stderr) for k, n in sorted(PATHS.items())])


class _Timed(object):
    """q: algorithmic bytes of the launch as this design runs it; q2 (default q): the COMPULSORY bytes -- without arrays the design
    writes for itself (the dense block's Hcat): what an honest roofline is priced against."""
    __slots__ = ('tag', 'q', 'f', 'q2', 'e0')

    def __init__(self, tag, q=0, f=0, q2=None):
        self.tag, self.q, self.f, self.q2 = tag, q, f, (q if q2 is None else q2)

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.setdefault(self.tag, []).append((self.e0, e1, self.q, self.f, self.q2))


def profile_summary(prof):
    """tag -> dict(launches, ms (mean per launch), bytes, flops, bytes_compulsory (means per launch))."""
    out = {}
    for tag, recs in prof.items():
        ms = [r[0].elapsed_time(r[1]) for r in recs]
        out[tag] = dict(launches=len(recs), ms=sum(ms) / len(ms), bytes=sum(r[2] for r in recs) / len(recs),
                        flops=sum(r[3] for r in recs) / len(recs), bytes_compulsory=sum(r[4] for r in recs) / len(recs))
    return out


def conv_cost(N, E, S, Fin, Fout):
    """fusion-agnostic compulsory traffic / flops of one SpectConv forward, fp32:
       Q = 4 (E S + N Fin + N Fout + S Fin Fout + Fout) + 4 (E + N + 1);  F = 2 E S Fin + 2 N S Fin Fout."""
    q = 4 * (E * S + N * Fin + N * Fout + S * Fin * Fout + Fout) + 4 * (E + N + 1)
    f = 2 * E * S * Fin + 2 * N * S * Fin * Fout
    return q, f


def conv_cost_bwd(N, E, S, Fin, Fout, need_x=True, need_val=True):
    """compulsory traffic / flops of one SpectConv backward incl. d/dval:
       Q = 4 (2 E S + 2 N Fin + N Fout + 2 S Fin Fout + Fout) + 4 (E + N + 1);  F = 6 E S Fin + 4 N S Fin Fout.
    A launch that produces no dX (the first layer: its input is data) is charged one N Fin (x is still read for dW and
    dval), one that produces no dval one E S: the bytes it actually has to move."""
    q = 4 * ((2 if need_val else 1) * E * S + (2 if need_x else 1) * N * Fin + N * Fout + 2 * S * Fin * Fout + Fout) + 4 * (E + N + 1)
    f = (6 if need_val else 4) * E * S * Fin + (4 if need_x else 2) * N * S * Fin * Fout
    return q, f


def _off(t, elems):
    return ctypes.c_void_p(t.data_ptr() + 4 * int(elems))


def _f32c(t, name):
    _require_cuda(t, name)
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32, got %s' % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _f32rows(t, name):
    """x as the kernels take it: float32 rows with unit column stride -- a row-strided view (a data set that hands out
    float4-addressable rows: dataset.DeviceDataset.batch_assembled) passes as it is, like the padded copy rows4 makes"""
    _require_cuda(t, name)
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32, got %s' % (name, t.dtype))
    return t if (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.size(1)) else t.contiguous()


# rows of x as the 8-wave kernels want them: float4-addressable (leading dimension a multiple of 4 floats, 16-byte aligned
# base), which is what the LDS-DMA landing ring of the forward (csrc/gml_spectconv_fwd3_impl.h) and the vector paths of the
# backward kernels copy.  Zinc12k.py's 21 + 4 = 25 input features (libs/utils.py:253-259) are not: the first layer's input is
# copied once into a zero-padded [N, 28] buffer; for a tensor that carries no gradient (per-batch data) the copy is kept per
# tensor object, so a batch that is stepped on repeatedly, or a data set that hands out padded rows itself, pays it once.
_ROWS4_CACHE = _collections.OrderedDict()


def rows4(x):
    if x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
        return x
    # While a HIP graph is being captured the copy must be RECORDED: a cache hit would leave it out of the graph, and every
    # replay after `static_x.copy_(new)` would read the padded copy of the warm-up data; the cached buffer could
    # also be evicted and freed while the graph still holds its address.  So: no lookup, no insert under capture.
    capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
    key = id(x)
    ent = None if capturing else _ROWS4_CACHE.get(key)
    if ent is not None and ent[0]() is x and ent[1] == x._version:
        _ROWS4_CACHE.move_to_end(key)
        return ent[2]
    F = int(x.size(1))
    buf = torch.zeros(x.size(0), (F + 3) // 4 * 4, dtype=x.dtype, device=x.device)
    buf[:, :F] = x
    v = buf[:, :F]
    if not x.requires_grad and not capturing:
        _ROWS4_CACHE[key] = (_weakref.ref(x), x._version, v)
        while len(_ROWS4_CACHE) > 4:
            _ROWS4_CACHE.popitem(last=False)
    return v


# The FORWARD projection of the 8-wave kernels (fwd3: every ML3Layer of the ZINC config; fwd2: counting.py's 12 supports; fwd4: sr25's / mutag's
# 48-wide layers) on f16 (hi, lo) pieces under power-of-two scales
# (GML_F16X3, csrc/gml_common.h "f16x3": residual 2^-24 per operand instead of 2^-17, same matrix-pipe instruction count) -- the
# default since round 6; GML_FWD_F16=0 restores the bf16 pairs.  With the three-piece edge forward (EDGE_FWD6) this makes the whole
# forward pass fp32-class, which is what trained-state gradients need (profiles/r06_precision_diag.jsonl); the backward kernels
# keep bf16x3 (measured not to matter).
FWD_F16 = _os.environ.get('GML_FWD_F16', '1') not in ('0', '')


def _fwd_arith_flags():
    return _lib.GML_F32_MFMA if exact_mode() else (_lib.GML_F16X3 if FWD_F16 else 0)


# ---------------------------------------------------------------------------- raw launches
def fused_conv(rowptr, col, ginfo, epos, val, x, ldx, w, w_strides, bias, out, ldo, nrows, S, Fin, Fout, flags=0,
               out_off=0, tag='spectconv_fwd'):
    q, f = conv_cost(int(nrows), int(val.size(0)), int(S), int(Fin), int(Fout)) if PROFILE is not None else (0, 0)
    with _Timed(tag, q + (4 * int(val.size(0)) if epos is not None else 0), f):
        _fused_conv(rowptr, col, ginfo, epos, val, x, ldx, w, w_strides, bias, out, ldo, nrows, S, Fin, Fout, flags,
                    out_off)


def _fused_conv(rowptr, col, ginfo, epos, val, x, ldx, w, w_strides, bias, out, ldo, nrows, S, Fin, Fout, flags,
                out_off):
    _lib.call('gml_spectconv_fwd', _ptr(rowptr), _ptr(col), _ptr(ginfo), _ptr(epos), _ptr(val), _ptr(x), int(ldx),
              _ptr(w), int(w_strides[0]), int(w_strides[1]), int(w_strides[2]), _ptr(bias),
              _off(out, out_off), int(ldo), int(nrows), int(S), int(Fin), int(Fout),
              int(flags) | _fwd_arith_flags(), _stream(x.device))


def conv_epilogue_applies(S, Fin, Fout):
    """shape class of gml_spectconv_fwd_epi (the ring kernel's ConCat / depthwise epilogues); GML_NO_EPILOGUE=1: the mapping"""
    return (not exact_mode()) and S in (4, 8) and Fin <= 32 and Fout <= 32 and not _os.environ.get('GML_NO_EPILOGUE') \
        and not _os.environ.get('GML_FWD64') and _os.environ.get('GML_FWD_DMA', '1') != '0'


def conv_epilogue(csr, x, val, w, bias, S, epilogue, ds, self_term, ncols):
    """one launch of gml_spectconv_fwd_epi: epilogue 1 = ConCat column blocks, 2 = depthwise; returns out [N, ncols] or None when
    the library reports the shape unsupported.  x: float4-addressable rows (rows4), val: target-sorted supports [E, S]."""
    Fin, Fout = int(w.size(-2)), int(w.size(-1))
    N = csr.N
    out = torch.empty(N, int(ncols), dtype=torch.float32, device=x.device)
    w = w.contiguous()
    b = _f32c(bias.detach(), 'bias') if bias is not None else None
    rc = _lib.lib().gml_spectconv_fwd_epi(_ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.ginfo128), _ptr(val), _ptr(x), int(x.stride(0)),
                                          _ptr(w), Fin * Fout, Fout, 1, _ptr(b), _ptr(out), int(ncols), N, int(S), Fin, Fout, 0,
                                          int(epilogue), _ptr(ds), 1 if self_term else 0, _stream(x.device))
    if rc == _lib.GML_E_UNSUPPORTED:
        return None
    _lib.check(rc)
    _path('conv_fwd', 'ring kernel, %s epilogue' % ('ConCat' if epilogue == 1 else 'depthwise'), S, Fin, Fout)
    return out


# Groups with more edges than one work item of the ring kernels stage at once (sr25.py: 13 entries per row) are walked in edge chunks
# on gml_k_spectconv_fwd4 (2-3 x the 64-row family's speed).  Opt-in through most of round 4 -- the chunked road differed from itself in
# ~1 launch of 200 in fresh processes -- until the cause was found (one v_pk_fma_f32 operand form, csrc/gml_spectconv_fwd4_impl.h `fma`,
# DESIGN s4.1c): 0 of 1536 since.  GML_FWD_CHUNKS=0 keeps such batches on the 64-row family.
FWD_CHUNKS = _os.environ.get('GML_FWD_CHUNKS', '1') not in ('0', '')


def fwd_groups(csr, x, S, Fin, Fout):
    """(group records, extra flags) the forward kernel wants for this shape: the 8-wave kernel on 128-row records when
    the shape is compiled for it, else the 64-row kernel."""
    flags = _lib.GML_F32_MFMA if exact_mode() else 0
    if _os.environ.get('GML_FWD64'):                         # experiments: force the 4-wave / 64-row kernel family
        return csr.ginfo, 0
    rows = int(_lib.lib().gml_spectconv_fwd_group_rows(int(S), int(Fin), int(Fout), flags))
    L = _lib.lib()
    gm_known = getattr(csr, 'gmax128', None) is not None
    gm_e, gm_w = csr.gmax128 if gm_known else (0, 0)
    x4 = x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
    # the chunked ring kernel (fwd4) addresses x with 32-bit byte offsets and needs the real group maxima to check its window: without
    # either, the 64-row family (any size, global gathers where a group does not fit) -- ADVICE r04
    small32 = (int(csr.N) + 16) * int(x.stride(0)) * 4 < 2 ** 31
    if rows == 128 and (S == 6 or (S == 4 and Fin > 32)) and not (gm_known and small32):
        rows = 64
    if rows == 128:
        cap = int(L.gml_spectconv_fwd_stage_edges(int(S), int(Fin), int(Fout), flags))      # edges of one work item of the ring kernel
        only4 = S == 6 or (S == 4 and Fin > 32)   # 6 supports / 48 features exist only on the chunked ring kernel: float4 rows, windows it can stage
        if only4:
            win = int(L.gml_spectconv_fwd_stage_window(int(S), int(Fin), int(Fout), flags))
            if not x4 or (win > 0 and gm_w > win) or (gm_e > cap and not FWD_CHUNKS):
                rows = 64
            elif gm_e > cap:
                # 48 features: one staged X window instead of two -> twice the edges per chunk (sr25: 2 chunks per group instead of 4)
                onewin = _lib.GML_FWD_ONEWIN if (Fin > 32 and _os.environ.get('GML_FWD_ONEWIN', '1') != '0') else 0
                _path('conv_fwd', 'fused 8-wave bf16x3, edge chunks%s' % (', one X window' if onewin else ''), S, Fin, Fout)
                return csr.ginfo128, _lib.GML_GROUPS128 | onewin
        elif cap > 0 and gm_e > cap and FWD_CHUNKS and x4 and small32:
            # a group beyond what the default ring kernel stages at once: its chunked form instead of global gathers
            win = int(L.gml_spectconv_fwd_stage_window(int(S), int(Fin), int(Fout), flags | _lib.GML_FWD_CHUNKED))
            if gm_w <= win:
                _path('conv_fwd', 'fused 8-wave bf16x3, edge chunks', S, Fin, Fout)
                return csr.ginfo128, _lib.GML_GROUPS128 | _lib.GML_FWD_CHUNKED
    if rows == 128:
        _path('conv_fwd', 'fused 8-wave bf16x3', S, Fin, Fout)
        return csr.ginfo128, _lib.GML_GROUPS128
    if rows == _lib.GML_GROUPS64_RANKED:
        if x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
            _path('conv_fwd', 'fused 4-wave geometry of the 8-wave kernel, 2 workgroups per CU, bf16x3', S, Fin, Fout)
            return csr.ranked64()[0], _lib.GML_GROUPS64R
        _path('conv_fwd', 'fused 8-wave bf16x3', S, Fin, Fout)
        return csr.ginfo128, _lib.GML_GROUPS128
    _path('conv_fwd', 'fused 4-wave (%s)' % ('f32 MFMA' if (exact_mode() or Fin <= 16 or Fout > 32) else 'bf16x3'), S, Fin, Fout)
    return csr.ginfo, 0


def fwd_gathers(S, Fin, Fout):
    """True when the forward of this shape runs on the 8-wave kernel, which can take its value rows through a position map
    (GML_EDGE_DUAL=1 keeps the round-1 scheme -- edge branch in target order, second copy scattered -- for A/B)."""
    if _os.environ.get('GML_FWD64') or _os.environ.get('GML_EDGE_DUAL'):
        return False
    flags = _lib.GML_F32_MFMA if exact_mode() else 0
    return int(_lib.lib().gml_spectconv_fwd_group_rows(int(S), int(Fin), int(Fout), flags)) in (128, _lib.GML_GROUPS64_RANKED)


def ml3_edge_in_source_order(csr, S, Fin, Fout):
    """True when a training ML3Layer of this shape runs its edge branch in source order (fused backward + gathering forward)."""
    return fused_bwd_available(csr, S, Fin, Fout) and fwd_gathers(S, Fin, Fout)


def spmm(csr, val, x, S, Fin):
    h = torch.empty(csr.N, S * Fin, dtype=torch.float32, device=x.device)
    # hint for the kernel choice: the largest 128-row group (of the source view -- SpectralDesign's masks are symmetric, so of
    # this view too; only the choice depends on it)
    gm = csr.gmax_t128[0] if getattr(csr, 'gmax_t128', None) is not None else -1
    _lib.call('gml_spmm_fwd_ex', _ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.ginfo128), _ptr(None), _ptr(val), _ptr(x), int(x.stride(0)),
              _ptr(h), csr.N, int(S), int(Fin), int(gm), _stream(x.device))
    return h


def sddmm(csr, x, gw, S, Fin):
    dval = torch.empty(csr.E, S, dtype=torch.float32, device=x.device)
    _lib.call('gml_sddmm', _ptr(csr.rowptr), _ptr(csr.col), _ptr(None), _ptr(x), int(x.stride(0)), _ptr(gw),
              _ptr(dval), csr.N, int(S), int(Fin), _stream(x.device))
    return dval


def relu_bwd(gy, gy_off, ldgy, y, ldy, nrows, F):
    """[nrows, F] view of a buffer whose rows are zero-padded to a multiple of 4 floats (float4-readable)."""
    ld = (F + 3) // 4 * 4
    g = torch.empty(nrows, ld, dtype=torch.float32, device=gy.device)
    _lib.call('gml_relu_bwd', _off(gy, gy_off), int(ldgy), _ptr(y), int(ldy), _ptr(g), ld, int(nrows), int(F),
              _stream(gy.device))
    return g[:, :F]


def edge_presplit(ea):
    """bf16 hi | lo image of the rows of ea for the matrix-core edge kernels (32 bytes per edge for S <= 8, 64 bytes for
    8 < S <= 16); None when not applicable."""
    E, S = ea.shape
    if S > 16 or E == 0:
        return None
    es = torch.empty(E, 8 if S <= 8 else 16, dtype=torch.int32, device=ea.device)
    _lib.call('gml_edge_presplit', _ptr(ea), _ptr(es), int(E), int(S), _stream(ea.device))
    return es


def _edge_mlp_pad(ea, w1, w2, w3, w4):
    """nedgeoutput != nedgeinput (spect_conv.py:66-71 allows it, no reference script uses it): the kernels are compiled
    for square shapes, and zero-padded operands of size P = max(S, Sout) give the identical result — a padded hidden
    unit is relu(0) = 0 resp. tanh(0)^2 = 0, a padded output column is dropped."""
    S, So = ea.size(1), w4.size(0)
    P = max(S, So)

    def pad_in(w):
        z = w.new_zeros(2 * P, P)
        z[:2 * S, :S] = w
        return z
    w4p = w4.new_zeros(P, 4 * P)
    w4p[:So, :2 * S] = w4[:, :2 * S]
    w4p[:So, 2 * P:2 * P + 2 * S] = w4[:, 2 * S:]
    return torch.nn.functional.pad(ea, (0, P - S)), pad_in(w1), pad_in(w2), pad_in(w3), w4p, P


def _edge_branch_torch(ea, w1, w2, w3, w4):
    """libs/spect_conv.py:205-207 as library calls (fc1_4 on the two halves of its input: no [E, 4S] concatenation)."""
    h1, h23 = torch.relu(_linear(ea, w1)), torch.tanh(_linear(ea, w2)) * torch.tanh(_linear(ea, w3))
    return torch.relu(torch.addmm(h1 @ w4[:, :h1.size(1)].t(), h23, w4[:, h1.size(1):].t()))


EDGE_WIDE_BWD_LIB = _os.environ.get('GML_EDGE_WIDE_BWD_LIB', '0') not in ('0', '')     # A/B: the round-5 library recompute


class EdgeBranchWide(torch.autograd.Function):
    """ML3Layer edge branch for 16 < max(S, Sout) <= 48: forward = ONE launch (gml_edge_mlp_wide_fwd: weights in LDS, exact fp32
    products, no intermediate in HBM -- the library road writes ~30 GB of them at S = 48 on 13 M edges); backward (round 6) = ONE launch
    for the per-edge part (gml_edge_mlp_wide_bwd: masked output gradient, hidden activations and the three pre-activation gradients,
    exact fp32) + the four weight gradients as tall contractions on the matrix cores (gml_xty_wide); the supports' own gradient, when
    asked for, is three library GEMMs.  GML_EDGE_WIDE_BWD_LIB=1: the library expression recomputed under autograd (rounds 1-5)."""

    @staticmethod
    def forward(ctx, ea, w1, w2, w3, w4):
        ea = _f32c(ea, 'edge_attr')
        ws = [_f32c(w.detach(), 'edge weight') for w in (w1, w2, w3, w4)]
        E, S = ea.shape
        So = int(w4.size(0))
        out = torch.empty(E, So, dtype=torch.float32, device=ea.device)
        with torch.cuda.device(ea.device):
            _lib.call('gml_edge_mlp_wide_fwd', _ptr(ea), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]), _ptr(ws[3]), _ptr(out), int(E), int(S), So,
                      _stream(ea.device))
        ctx.save_for_backward(ea, w1, w2, w3, w4, out)
        return out

    @staticmethod
    def backward(ctx, g):
        ea, w1, w2, w3, w4, out = ctx.saved_tensors
        need = ctx.needs_input_grad
        E, S = ea.shape
        So = int(w4.size(0))
        if not EDGE_WIDE_BWD_LIB and not exact_mode('edge') and E > 0:
            dev = ea.device
            ws = [_f32c(w.detach(), 'edge weight') for w in (w1, w2, w3, w4)]
            g = _f32c(g, 'grad_output')
            H2 = 2 * S
            H2R = int(_lib.lib().gml_edge_mlp_wide_bwd_h2r(int(S)))
            with torch.cuda.device(dev):
                go = torch.empty(E, So, dtype=torch.float32, device=dev)
                hid = torch.empty(E, 2 * H2R, dtype=torch.float32, device=dev)
                gz = torch.empty(E, 3 * H2R, dtype=torch.float32, device=dev)
                _lib.call('gml_edge_mlp_wide_bwd', _ptr(ea), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]), _ptr(ws[3]), _ptr(out), _ptr(g), _ptr(go),
                          _ptr(hid), _ptr(gz), int(E), int(S), So, _stream(dev))
                dws = [xty_wide(gz[:, m * H2R:m * H2R + H2], ea) for m in range(3)]           # dW_m = gz_m^T e   [2S, S]
                d4 = [xty_wide(hid[:, b * H2R:b * H2R + H2], go) for b in range(2)]           # (hid_b^T go)^T    [So, 2S] each
                if all(t is not None for t in dws + d4):
                    dw4 = torch.cat([d4[0].t(), d4[1].t()], 1).contiguous()
                    gin = None
                    if need[0]:                            # d e = sum_m gz_m W_m (no reference script trains the raw supports)
                        gin = gz[:, :H2] @ ws[0]
                        gin.addmm_(gz[:, H2R:H2R + H2], ws[1]).addmm_(gz[:, 2 * H2R:2 * H2R + H2], ws[2])
                    return gin, dws[0], dws[1], dws[2], dw4
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(n) for t, n in zip((ea, w1, w2, w3, w4), need)]
            o2 = _edge_branch_torch(*ins)
            wanted = [t for t, n in zip(ins, need) if n]
            gs = iter(torch.autograd.grad(o2, wanted, g) if wanted else ())
        return tuple(next(gs) if n else None for n in need)


def edge_mlp_fwd(ea, w1, w2, w3, w4, tpos=None, ea_split=None):
    """returns out (same edge order as ea) and, when tpos is given, the same rows at out_t[tpos[e]]."""
    E, S = ea.shape
    So = w4.size(0)
    if So != S:
        eap, w1p, w2p, w3p, w4p, _ = _edge_mlp_pad(ea, w1, w2, w3, w4)
        out, out_t = edge_mlp_fwd(eap, w1p, w2p, w3p, w4p, tpos, edge_presplit(eap))
        return out[:, :So].contiguous(), (out_t[:, :So].contiguous() if out_t is not None else None)
    out = torch.empty(E, So, dtype=torch.float32, device=ea.device)
    out_t = torch.empty(E, So, dtype=torch.float32, device=ea.device) if tpos is not None else None
    if exact_mode('edge'):                                # exact arithmetic covers the edge branch too (round 5: it did not)
        _lib.call('gml_edge_mlp_fwd_exact', _ptr(ea), _ptr(w1), _ptr(w2), _ptr(w3), _ptr(w4), _ptr(out), _ptr(tpos), _ptr(out_t),
                  int(E), int(S), int(So), _stream(ea.device))
        return out, out_t
    if EDGE_FWD6 and 2 <= S <= 16 and not EDGE_VALU:      # three-piece products (fp32-class), reads the fp32 rows itself
        rc = _lib.lib().gml_edge_mlp_fwd6(_ptr(ea), _ptr(w1), _ptr(w2), _ptr(w3), _ptr(w4), _ptr(out), _ptr(tpos), _ptr(out_t),
                                          int(E), int(S), int(So), _stream(ea.device))
        if rc != _lib.GML_E_UNSUPPORTED:
            _lib.check(rc)
            return out, out_t
    _lib.call('gml_edge_mlp_fwd', _ptr(ea), _ptr(ea_split), _ptr(w1), _ptr(w2), _ptr(w3), _ptr(w4), _ptr(out), _ptr(tpos),
              _ptr(out_t), int(E), int(S), int(So), _stream(ea.device))
    return out, out_t


EDGE_STACK = not _os.environ.get('GML_NO_EDGE_STACK')     # A/B switch: edge branches of stacked layers in one launch
# The edge-branch FORWARD on three-piece products ("bf16x6", csrc/gml_edge_chain6_impl.h: fp32-class learned supports) -- the default
# since round 6; GML_EDGE_FWD6=0 restores the two-piece chain (bf16x3: ~5e-7 rms on the supports, which after training moved
# parameter gradients to 1e-3 .. 1e-2 of their term sums: profiles/r06_precision_diag.jsonl).
EDGE_FWD6 = _os.environ.get('GML_EDGE_FWD6', '1') not in ('0', '')
# The edge branch over a batch's UNIQUE support rows (csrc/gml_edge_chain_sym_impl.h: an edge and its mirror mostly carry bitwise the
# same row -- the supports sample symmetric matrices -- and then get the same output; exact, no tolerance): GML_EDGE_SYM=0 turns it off.
EDGE_SYM = _os.environ.get('GML_EDGE_SYM', '1') not in ('0', '')


def edge_mlp_fwd_stack(ea, ea_split, weights, sym=None):
    """The edge branches of several layers on the SAME supports in one pass (gml_edge_mlp_fwd_stack): weights = [(w1, w2, w3, w4)]
    per layer; returns the list of outputs [E, S] (edge order of ea), or None when the library has no stacked kernel for the shape."""
    import ctypes
    E, S = ea.shape
    L = len(weights)
    if exact_mode('edge') or not ((1 if sym is not None else 2) <= L <= 4) or any(w[3].size(0) != S or w[0].size(1) != S for w in weights):
        return None                                       # (the stacked kernel is a matrix-core chain: not the exact arithmetic)
    arr = lambda ts: (ctypes.c_void_p * L)(*[t.data_ptr() for t in ts])
    if EDGE_FWD6 and not EDGE_VALU and (S in (4, 8) or (L == 1 and 2 <= S <= 16)) and sym is not None and EDGE_SYM:
        # the unique support rows only (gml_edge_chain_sym_impl.h): every output row is written, by its own entry or by its mirror's
        outs = [torch.empty(E, S, dtype=torch.float32, device=ea.device) for _ in range(L)]
        rc = _lib.lib().gml_edge_mlp_fwd_stack6_sym(_ptr(ea), _ptr(sym[0]), _ptr(sym[1]), int(sym[0].numel()), L, arr([w[0] for w in weights]),
                                                    arr([w[1] for w in weights]), arr([w[2] for w in weights]), arr([w[3] for w in weights]),
                                                    arr(outs), int(E), int(S), int(S), _stream(ea.device))
        if rc != _lib.GML_E_UNSUPPORTED:
            _lib.check(rc)
            return outs
    if EDGE_FWD6 and not EDGE_VALU and S in (4, 8):
        outs = [torch.empty(E, S, dtype=torch.float32, device=ea.device) for _ in range(L)]
        rc = _lib.lib().gml_edge_mlp_fwd_stack6(_ptr(ea), L, arr([w[0] for w in weights]), arr([w[1] for w in weights]),
                                                arr([w[2] for w in weights]), arr([w[3] for w in weights]), arr(outs), int(E), int(S),
                                                int(S), _stream(ea.device))
        if rc != _lib.GML_E_UNSUPPORTED:
            _lib.check(rc)
            return outs
    if ea_split is None:
        return None
    outs = [torch.empty(E, S, dtype=torch.float32, device=ea.device) for _ in range(L)]
    rc = _lib.lib().gml_edge_mlp_fwd_stack(_ptr(ea_split), L, arr([w[0] for w in weights]), arr([w[1] for w in weights]),
                                           arr([w[2] for w in weights]), arr([w[3] for w in weights]), arr(outs), int(E), int(S),
                                           int(S), _stream(ea.device))
    if rc == _lib.GML_E_UNSUPPORTED:
        return None
    _lib.check(rc)
    return outs


def edge_mlp_bwd(ea, w1, w2, w3, w4, gout, need_gin, ea_split=None, sym=None):
    E, S = ea.shape
    So = w4.size(0)
    dev = ea.device
    if So != S:
        eap, w1p, w2p, w3p, w4p, P = _edge_mlp_pad(ea, w1, w2, w3, w4)
        gin, d1, d2, d3, d4 = edge_mlp_bwd(eap, w1p, w2p, w3p, w4p, torch.nn.functional.pad(gout, (0, P - So)), need_gin,
                                           edge_presplit(eap))
        return (gin[:, :S].contiguous() if gin is not None else None, d1[:2 * S, :S].contiguous(),
                d2[:2 * S, :S].contiguous(), d3[:2 * S, :S].contiguous(),
                torch.cat([d4[:So, :2 * S], d4[:So, 2 * P:2 * P + 2 * S]], 1))
    nbytes = int(_lib.lib().gml_edge_mlp_bwd_workspace_bytes(int(E), int(S), int(So)))
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
    gin = torch.empty_like(ea) if need_gin else None
    dw1, dw2, dw3, dw4 = torch.empty_like(w1), torch.empty_like(w2), torch.empty_like(w3), torch.empty_like(w4)
    if exact_mode('edge'):
        _lib.call('gml_edge_mlp_bwd_exact', _ptr(ea), _ptr(w1), _ptr(w2), _ptr(w3), _ptr(w4), _ptr(gout), _ptr(gin),
                  _ptr(dw1), _ptr(dw2), _ptr(dw3), _ptr(dw4), int(E), int(S), int(So), _ptr(ws), ws.numel(), _stream(dev))
        return gin, dw1, dw2, dw3, dw4
    fq = _fold_queue()
    if sym is not None and EDGE_SYM and not need_gin and ea_split is not None and 2 <= S <= 16 and not EDGE_VALU and E > 0:
        # unique support rows only: entry u runs the chain once on gout[uid[u]] + gout[mir[u]]
        U = int(sym[0].numel())
        nofold = fq is not None
        d = [_ptr(None)] * 4 if nofold else [_ptr(dw1), _ptr(dw2), _ptr(dw3), _ptr(dw4)]
        rc = _lib.lib().gml_edge_mlp_bwd_sym(_ptr(ea_split), _ptr(sym[0]), _ptr(sym[1]), U, _ptr(w1), _ptr(w2), _ptr(w3), _ptr(w4), _ptr(gout),
                                             d[0], d[1], d[2], d[3], int(E), int(S), int(So), _ptr(ws), ws.numel(), _stream(dev))
        if rc != _lib.GML_E_UNSUPPORTED:
            _lib.check(rc)
            if not nofold:
                return None, dw1, dw2, dw3, dw4
            n1, n4 = 2 * S * S, 4 * S * So
            flat = torch.empty(3 * n1 + n4, dtype=torch.float32, device=dev)
            fq.append((ws, int(_lib.lib().gml_edge_mlp_bwd_sym_parts(U, int(S))), 3 * n1 + n4, [(flat, 3 * n1 + n4)]))
            return (None, flat[:n1].view_as(w1), flat[n1:2 * n1].view_as(w2), flat[2 * n1:3 * n1].view_as(w3), flat[3 * n1:].view_as(w4))
    if fq is not None and E > 0:
        nparts = int(_lib.lib().gml_edge_mlp_bwd_parts(int(E), int(S), int(So), 1 if ea_split is not None else 0, 1 if need_gin else 0))
        _lib.call('gml_edge_mlp_bwd', _ptr(ea), _ptr(ea_split), _ptr(w1), _ptr(w2), _ptr(w3), _ptr(w4), _ptr(gout), _ptr(gin),
                  _ptr(None), _ptr(None), _ptr(None), _ptr(None), int(E), int(S), int(So), _ptr(ws), ws.numel(), _stream(dev))
        # ONE flat destination, the gradients are views of it: autograd takes a view over as .grad without a copy, and the flat
        # tensor (held by the queue until the fold has run) cannot be freed under the fold's feet -- separate tensors referenced by
        # the queue would be CLONED by AccumulateGrad (a second owner), before the fold has filled them
        n1, n4 = 2 * S * S, 4 * S * So
        flat = torch.empty(3 * n1 + n4, dtype=torch.float32, device=dev)
        fq.append((ws, nparts, 3 * n1 + n4, [(flat, 3 * n1 + n4)]))
        return (gin, flat[:n1].view_as(w1), flat[n1:2 * n1].view_as(w2), flat[2 * n1:3 * n1].view_as(w3), flat[3 * n1:].view_as(w4))
    _lib.call('gml_edge_mlp_bwd', _ptr(ea), _ptr(ea_split), _ptr(w1), _ptr(w2), _ptr(w3), _ptr(w4), _ptr(gout), _ptr(gin),
              _ptr(dw1), _ptr(dw2), _ptr(dw3), _ptr(dw4), int(E), int(S), int(So), _ptr(ws), ws.numel(), _stream(dev))
    return gin, dw1, dw2, dw3, dw4


def node_mix_native(Fin, F2):
    """True when the Hadamard branch tanh(fc11 x) * tanh(fc12 x) of this shape runs on the fused HIP kernels."""
    return int(_lib.lib().gml_node_mix_bwd_workspace_bytes(1, int(Fin), int(F2))) != 0


def ml3_split_bwd(gy, y, nout1, x=None, w11=None, b11=None, w12=None, b12=None, need_dx=False, need_dcb=False, dz_out=False,
                  gy_seg=None, premasked=False):
    """One pass over the rows for the backward of  cat[relu(conv), tanh(fc11 x) * tanh(fc12 x)]  (or of a plain
    relu(conv) when w11 is None): returns G [N, nout1] (view of a zero-padded buffer), dx (Hadamard-branch part
    only, or None), dcb, dw11, db11, dw12, db12.  dz_out (2 nout2 <= 4): the second result is dz [N, 4] = (dz11 | dz12)
    instead of dx = dz [w11; w12] (the operand of fused_conv_bwd(..., mix=)).  gy_seg (int32 [N]): gy has one row per
    segment and row r reads gy[gy_seg[r]] (the un-expanded gradient of a global add pool that follows the layer).
    premasked: gy[:, :nout1] already carries this layer's relu mask (the consumer layer's conv backward applied it,
    fused_conv_bwd(..., relu_cols=)): y is not read, no G is written -- the returned G is a view of gy."""
    N = y.size(0)
    F2 = 0 if w11 is None else int(w11.size(0))
    Fin = int(x.size(1)) if F2 else 0
    dev = gy.device
    nbytes = int(_lib.lib().gml_ml3_split_bwd_workspace_bytes(int(N), Fin, int(nout1), F2))
    if nbytes == 0 and N > 0:
        return None
    ld = (nout1 + 3) // 4 * 4                                # zero padded: float4-readable rows
    premasked = bool(premasked and gy_seg is None and gy.stride(0) % 4 == 0 and gy.stride(0) >= ld and gy.data_ptr() % 16 == 0)
    G = gy if premasked else torch.empty(N, ld, dtype=torch.float32, device=dev)
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
    dx = torch.empty(N, Fin, dtype=torch.float32, device=dev) if (need_dx and F2 and not dz_out) else None
    dz = torch.empty(N, 4, dtype=torch.float32, device=dev) if (dz_out and F2) else None
    dcb = torch.empty(nout1, dtype=torch.float32, device=dev) if need_dcb else None
    dw11 = torch.empty_like(w11) if F2 else None
    dw12 = torch.empty_like(w12) if F2 else None
    db11 = torch.empty_like(b11) if (F2 and b11 is not None) else None
    db12 = torch.empty_like(b12) if (F2 and b12 is not None) else None
    fq = _fold_queue()
    kd = (dcb, dw11, db11, dw12, db12)                       # what the kernel's own fold writes
    if fq is not None and N > 0 and any(t is not None for t in kd):
        # deferred: no destinations -> the entry point skips its fold; the partial rows [dw11 | dw12 | db11 | db12 | dcb] stay in ws
        npart = 2 * F2 * Fin + 2 * F2 + int(nout1)
        flat = torch.empty(npart, dtype=torch.float32, device=dev)                 # (views of it go to autograd: see edge_mlp_bwd)
        fq.append((ws, nbytes // (4 * npart), npart, [(flat, npart)]))
        o = 0
        if F2:
            dw11, dw12 = flat[:F2 * Fin].view_as(w11), flat[F2 * Fin:2 * F2 * Fin].view_as(w12)
            o = 2 * F2 * Fin
            db11 = flat[o:o + F2] if db11 is not None else None
            db12 = flat[o + F2:o + 2 * F2] if db12 is not None else None
            o += 2 * F2
        dcb = flat[o:o + int(nout1)] if dcb is not None else None
        kd = (None,) * 5
    kcb, k11, kb11, k12, kb12 = kd
    with _Timed('ml3_split_bwd', 4 * N * ((1 if premasked else 3) * (nout1 + F2) + (2 * Fin if dx is not None else Fin) + (4 if dz is not None else 0))):
        if dz is not None or gy_seg is not None or premasked:
            _lib.call('gml_ml3_split_bwd_ex', _ptr(gy), int(gy.stride(0)), _ptr(gy_seg), _ptr(None if premasked else y), int(y.stride(0)), _ptr(x),
                      int(x.stride(0)) if F2 else 0, _ptr(w11), _ptr(b11), _ptr(w12), _ptr(b12), _ptr(None if premasked else G), ld, _ptr(dx), Fin,
                      _ptr(dz), _ptr(kcb), _ptr(k11), _ptr(kb11), _ptr(k12), _ptr(kb12), int(N), Fin, int(nout1), F2,
                      _ptr(ws), ws.numel(), _stream(dev))
        else:
            _lib.call('gml_ml3_split_bwd', _ptr(gy), int(gy.stride(0)), _ptr(y), int(y.stride(0)), _ptr(x),
                      int(x.stride(0)) if F2 else 0, _ptr(w11), _ptr(b11), _ptr(w12), _ptr(b12), _ptr(G), ld, _ptr(dx), Fin,
                      _ptr(kcb), _ptr(k11), _ptr(kb11), _ptr(k12), _ptr(kb12), int(N), Fin, int(nout1), F2, _ptr(ws),
                      ws.numel(), _stream(dev))
    return G[:, :nout1], (dz if dz is not None else dx), dcb, dw11, db11, dw12, db12


def xty(a, b):
    """a^T b for two tall matrices with <= 64 columns each ([n,p], [n,q] -> [p,q]); None if outside the kernel range."""
    n, p, q = int(a.size(0)), int(a.size(1)), int(b.size(1))
    nbytes = int(_lib.lib().gml_xty_workspace_bytes(n, p, q))
    if nbytes == 0 and n > 0:
        return None
    out = torch.empty(p, q, dtype=torch.float32, device=a.device)
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=a.device)
    with _Timed('xty'):
        _lib.call('gml_xty', _ptr(a), int(a.stride(0)), _ptr(b), int(b.stride(0)), _ptr(out), n, p, q, _ptr(ws), ws.numel(),
                  _stream(a.device))
    return out


def gnnml1_block_supported(x, Fin, n1, n2, n3, mode):
    return (x.is_cuda and x.dtype == torch.float32 and not _os.environ.get('GML_NO_GNNML1_FUSED')
            and bool(_lib.lib().gml_gnnml1_supported(int(Fin), int(n1), int(n2), int(n3), int(mode))))


class GNNML1BlockFunction(torch.autograd.Function):
    """One GNNML1 block (sr25.py:231-240, mnist75.py:296-318, mutag.py:253-262) as ONE launch forward and one launch + four gml_xty
    backward (csrc/gml_gnnml1.hip): a = fc_i1 x, c = conv_i1 x (SpectConv K = 1), f2 = fc_i2 x, f3 = fc_i3 x;
    mode 0: act(a + c + f2 f3); 1: [act a | act c | act(f2 f3)]; 2: [act a | act c | act f2 . act f3]; act 0 tanh / 1 relu.
    val: per-edge values in TARGET order ([E] / [E, 1]) or None for ones (the scripts pass torch.ones); they carry no gradient here
    (the module takes the unfused road when edge_attr requires one).  Exact fp32 products."""

    @staticmethod
    def forward(ctx, x, csr, val, w1, b1, wc, bc, w2, b2, w3, b3, mode, act):
        x = _f32rows(x, 'x')
        N, Fin = int(x.size(0)), int(x.size(1))
        n1, n2, n3 = int(w1.size(0)), int(wc.size(-1)), int(w2.size(0))
        w1, w2, w3, wc = _f32c(w1, 'fc1.weight'), _f32c(w2, 'fc2.weight'), _f32c(w3, 'fc3.weight'), _f32c(wc, 'conv.weight')
        C = n1 if mode == 0 else n1 + n2 + n3
        dev = x.device
        if val is not None:
            val = _f32c(val.reshape(-1), 'edge_attr')
        with torch.cuda.device(dev):
            out = torch.empty(N, C, dtype=torch.float32, device=dev)
            with _Timed('gnnml1_fwd', 4 * (N * (Fin + C) + csr.E + N + Fin * (n1 + n2 + 2 * n3)) if PROFILE is not None else 0, 0):
                _lib.call('gml_gnnml1_fwd', _ptr(csr.rowptr), _ptr(csr.col), _ptr(val), _ptr(x), int(x.stride(0)), N, Fin,
                          _ptr(w1), _ptr(b1), n1, _ptr(wc), _ptr(bc), n2, _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), n3, int(mode), int(act),
                          _ptr(out), C, _stream(dev))
        ctx.csr, ctx.mode, ctx.act, ctx.dims = csr, int(mode), int(act), (N, Fin, n1, n2, n3, C)
        ctx.has_b = (b1 is not None, bc is not None, b2 is not None, b3 is not None)
        ctx.save_for_backward(x, val, w1, wc, w2, b2, w3, b3, out)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, val, w1, wc, w2, b2, w3, b3, out = ctx.saved_tensors
        N, Fin, n1, n2, n3, C = ctx.dims
        csr, mode, act = ctx.csr, ctx.mode, ctx.act
        dev = x.device
        gout = _f32rows(gout, 'grad_output')
        L = _lib.lib()
        ng4 = int(L.gml_gnnml1_g4_cols(n1, n2, n3, mode))
        p1, p2, p3 = (n1 + 15) // 16 * 16, (n2 + 15) // 16 * 16, (n3 + 15) // 16 * 16
        need_x = ctx.needs_input_grad[0]
        with torch.cuda.device(dev):
            val_t = csr.to_source_order(val.view(-1, 1)).view(-1) if val is not None else None
            g4 = torch.empty(N, ng4, dtype=torch.float32, device=dev)
            q = torch.empty(N, p2, dtype=torch.float32, device=dev)
            dx = torch.empty(N, Fin, dtype=torch.float32, device=dev) if need_x else None
            with _Timed('gnnml1_bwd', 4 * (N * (Fin + 2 * C + (Fin if need_x else 0) + ng4 + p2) + csr.E + N) if PROFILE is not None else 0, 0):
                _lib.call('gml_gnnml1_bwd', _ptr(csr.rowptr_t), _ptr(csr.col_t), _ptr(val_t), _ptr(x), int(x.stride(0)), _ptr(out), C,
                          _ptr(gout), int(gout.stride(0)), N, Fin, _ptr(w1), n1, _ptr(wc), n2, _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), n3,
                          mode, act, _ptr(dx), Fin, _ptr(g4), ng4, _ptr(q), p2, _stream(dev))
            oa, oc = 0, p1
            o2 = p1 if mode == 0 else p1 + p2
            o3 = o2 + p3
            with _Timed('gnnml1_dw'):
                nflat = int(L.gml_gnnml1_dw_floats(Fin, n1, n2, n3, mode))
                nws = int(L.gml_gnnml1_dw_workspace_bytes(N, Fin, n1, n2, n3, mode))
                flat = torch.empty(nflat, dtype=torch.float32, device=dev)
                ws = torch.empty(max(nws, 4), dtype=torch.uint8, device=dev)
                _lib.call('gml_gnnml1_dw', _ptr(x), int(x.stride(0)), _ptr(g4), ng4, _ptr(q), p2, N, Fin, n1, n2, n3, mode, _ptr(flat),
                          _ptr(ws), ws.numel(), _stream(dev))
                e1, e2, e3, e4 = n1 * Fin, n1 * Fin + n3 * Fin, n1 * Fin + 2 * n3 * Fin, n1 * Fin + 2 * n3 * Fin + Fin * n2
                dw1, dw2, dw3, dwc = flat[:e1].view(n1, Fin), flat[e1:e2].view(n3, Fin), flat[e2:e3].view(n3, Fin), flat[e3:e4].view(Fin, n2)
                sums = flat[e4:]
        hb1, hbc, hb2, hb3 = ctx.has_b
        db1 = sums[oa:oa + n1] if hb1 else None
        # mode 0: fc_i1's and conv_i1's biases receive the SAME column sums -- as two tensors with their own memory (autograd adopts a
        # returned gradient as .grad; two parameters sharing one buffer would double every in-place op on the gradients: ADVICE r05)
        dbc = ((sums[oa:oa + n2].clone() if hb1 else sums[oa:oa + n2]) if mode == 0 else sums[oc:oc + n2]) if hbc else None
        db2 = sums[o2:o2 + n3] if hb2 else None
        db3 = sums[o3:o3 + n3] if hb3 else None
        return dx, None, None, dw1, db1, dwc.view(1, Fin, n2), dbc, dw2, db2, dw3, db3, None, None


def xty_wide(a, b):
    """a^T b for a tall wide a [n, p <= 4096] and b [n, q <= 128] on the bf16 matrix cores (bf16x3 products; gml_xty_wide): the dense
    block's dW = Hcat^T g; None if outside the kernel range."""
    n, p, q = int(a.size(0)), int(a.size(1)), int(b.size(1))
    L = _lib.lib()
    if not L.gml_xty_wide_supported(n, p, q) or exact_mode() or a.stride(1) != 1 or b.stride(1) != 1:
        return None
    nbytes = int(L.gml_xty_wide_workspace_bytes(n, p, q))
    out = torch.empty(p, q, dtype=torch.float32, device=a.device)
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=a.device)
    _lib.call('gml_xty_wide', _ptr(a), int(a.stride(0)), _ptr(b), int(b.stride(0)), _ptr(out), n, p, q, _ptr(ws), ws.numel(),
              _stream(a.device))
    return out


class BatchNormFunction(torch.autograd.Function):
    """torch.nn.BatchNorm1d in training mode on the HIP kernels of csrc/gml_bn.hip (mutag.py:272-288: BatchNorm between the layers).
    Returns (y, batch mean, biased batch variance); raises NotImplementedError for shapes the kernels do not take (the module then
    uses torch's implementation)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        N, C = int(x.size(0)), int(x.size(1))
        dev = x.device
        L = _lib.lib()
        with torch.cuda.device(dev):
            st = _stream(dev)
            stats = torch.empty(3, C, dtype=torch.float32, device=dev)
            ws = torch.empty(max(int(L.gml_bn_workspace_bytes(N)), 4), dtype=torch.uint8, device=dev)
            with _Timed('bn_fwd'):
                rc = L.gml_bn_stats(_ptr(x), int(x.stride(0)), N, C, float(eps), _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(ws), ws.numel(), st)
            if rc == _lib.GML_E_UNSUPPORTED:
                raise NotImplementedError('shape')
            _lib.check(rc)
            y = torch.empty(N, C, dtype=torch.float32, device=dev)
            with _Timed('bn_fwd'):
                _lib.call('gml_bn_apply', _ptr(x), int(x.stride(0)), N, C, _ptr(stats[0]), _ptr(stats[2]), _ptr(weight), _ptr(bias), _ptr(y), C, st)
        ctx.save_for_backward(x, weight, stats)
        ctx.has_bias = bias is not None
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _ds):
        x, weight, stats = ctx.saved_tensors
        N, C = int(x.size(0)), int(x.size(1))
        dev = x.device
        dy = dy if (dy.stride(1) == 1 and dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0) else dy.contiguous()
        with torch.cuda.device(dev):
            st = _stream(dev)
            sums = torch.empty(2, C, dtype=torch.float32, device=dev)
            ws = torch.empty(max(int(_lib.lib().gml_bn_workspace_bytes(N)), 4), dtype=torch.uint8, device=dev)
            with _Timed('bn_bwd'):
                _lib.call('gml_bn_bwd_sums', _ptr(dy), int(dy.stride(0)), _ptr(x), int(x.stride(0)), N, C, _ptr(stats[0]), _ptr(stats[2]),
                          _ptr(sums[0]), _ptr(sums[1]), _ptr(ws), ws.numel(), st)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty(N, C, dtype=torch.float32, device=dev)
                with _Timed('bn_bwd'):
                    _lib.call('gml_bn_bwd_apply', _ptr(dy), int(dy.stride(0)), _ptr(x), int(x.stride(0)), N, C, _ptr(stats[0]), _ptr(stats[2]),
                              _ptr(weight), _ptr(sums[0]), _ptr(sums[1]), _ptr(dx), C, st)
        return dx, (sums[1] if weight is not None else None), (sums[0] if ctx.has_bias else None), None


class TallLinearFunction(torch.autograd.Function):
    """F.linear for a small layer applied to many rows; only the weight gradient g^T x differs from autograd's
    (a K = rows contraction that a library GEMM runs on a single workgroup)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = g.mm(w) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            with torch.cuda.device(x.device):
                # (a few rows -- the reference's batch 64: 65 pooled rows -- are one short library GEMM: 5 us against the 17 us the
                #  tall-matrix kernel's two stages take on nothing)
                gw = xty(g, x.contiguous()) if (x.is_cuda and x.dtype == torch.float32 and x.size(0) >= 2048) else None
            if gw is None:
                gw = g.t().mm(x)
        gb = g.sum(0) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


class HeadL1Function(torch.autograd.Function):
    """loss = sum_{r < len(y)} valid[r] |fc2(relu(fc1 p[r])) - y[r]| (Zinc12k.py:343-345, :365) as ONE launch forward and ONE launch
    backward (csrc/gml_head.hip) for small batches -- the reference's batch 64.  p [R, nin] pooled rows (R >= len(y): further rows,
    e.g. the padding graph of a static batch, are ignored and get zero gradient)."""

    @staticmethod
    def forward(ctx, p, y, valid, w1, b1, w2, b2, loss_sum=None):
        p = _f32c(p, 'pooled')
        R, nin = p.shape
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            # loss_sum (optional, a float32 scalar on the device): += loss inside the same launch -- an epoch's running loss
            with _Timed('head'):
                _lib.call('gml_head_l1_fwd_acc', _ptr(p), int(p.stride(0)), _ptr(y), _ptr(valid), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2),
                          int(R), int(y.numel()), int(nin), int(w1.size(0)), _ptr(loss), _ptr(loss_sum), _ptr(None), _stream(p.device))
        ctx.save_for_backward(p, y, valid, w1, b1, w2, b2)
        return loss

    @staticmethod
    def backward(ctx, g):
        p, y, valid, w1, b1, w2, b2 = ctx.saved_tensors
        R, nin = p.shape
        nh = int(w1.size(0))
        gp = torch.empty(R, nin, dtype=torch.float32, device=p.device)
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        db1 = torch.empty_like(b1) if b1 is not None else None
        db2 = torch.empty_like(b2) if b2 is not None else None
        g = g.contiguous()
        with torch.cuda.device(p.device):
            with _Timed('head'):
                _lib.call('gml_head_l1_bwd', _ptr(p), int(p.stride(0)), _ptr(y), _ptr(valid), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2),
                          int(R), int(y.numel()), int(nin), nh, _ptr(g), _ptr(gp), nin, _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2),
                          _stream(p.device))
        return gp, None, None, dw1, db1, dw2, db2, None


class HeadL1BigFunction(torch.autograd.Function):
    """HeadL1Function for any number of rows (nin = nh = 32; csrc/gml_head_big.hip): one pass + a fold each way instead of the ~16
    launches of the general road (two library GEMMs, relu, the loss's elementwise chain and their backward) at the bench's batch size."""

    @staticmethod
    def forward(ctx, p, y, valid, w1, b1, w2, b2, loss_sum=None):
        p = _f32c(p, 'pooled')
        R, nin = p.shape
        nws = int(_lib.lib().gml_head_l1_big_workspace_floats(int(R), int(nin), int(w1.size(0))))
        ws = torch.empty(nws, dtype=torch.float32, device=p.device)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            with _Timed('head'):
                _lib.call('gml_head_l1_big_fwd', _ptr(p), int(p.stride(0)), _ptr(y), _ptr(valid), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2),
                          int(R), int(y.numel()), int(nin), int(w1.size(0)), _ptr(loss), _ptr(loss_sum), _ptr(ws), nws, _stream(p.device))
        ctx.save_for_backward(p, y, valid, w1, b1, w2, b2)
        ctx.nws = nws
        return loss

    @staticmethod
    def backward(ctx, g):
        p, y, valid, w1, b1, w2, b2 = ctx.saved_tensors
        R, nin = p.shape
        nh = int(w1.size(0))
        dev = p.device
        gp = torch.empty(R, nin, dtype=torch.float32, device=dev)
        ws = torch.empty(ctx.nws, dtype=torch.float32, device=dev)
        g = g.contiguous()
        npart = nh * nin + 2 * nh + 1
        fq = _fold_queue()
        if fq is not None:                                   # the partials stay in ws; the gradients are views of the flat buffer the fold fills
            flat = torch.empty(npart, dtype=torch.float32, device=dev)
            fq.append((ws, ctx.nws // npart, npart, [(flat, npart)]))
            dw1, db1 = flat[:nh * nin].view_as(w1), (flat[nh * nin:nh * nin + nh] if b1 is not None else None)
            dw2 = flat[nh * nin + nh:nh * nin + 2 * nh].view_as(w2)
            db2 = flat[nh * nin + 2 * nh:] if b2 is not None else None
            kd = (None,) * 4
        else:
            dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
            db1 = torch.empty_like(b1) if b1 is not None else None
            db2 = torch.empty_like(b2) if b2 is not None else None
            kd = (dw1, db1, dw2, db2)
        with torch.cuda.device(dev):
            with _Timed('head'):
                _lib.call('gml_head_l1_big_bwd', _ptr(p), int(p.stride(0)), _ptr(y), _ptr(valid), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2),
                          int(R), int(y.numel()), int(nin), nh, _ptr(g), _ptr(gp), nin, _ptr(kd[0]), _ptr(kd[1]), _ptr(kd[2]), _ptr(kd[3]),
                          _ptr(ws), ctx.nws, _stream(dev))
        return gp, None, None, dw1, db1, dw2, db2, None


def head_l1_big_supported(p, w1, w2):
    return (p.is_cuda and p.dtype == torch.float32 and p.dim() == 2 and int(p.size(1)) == 32 and tuple(w1.shape) == (32, 32)
            and tuple(w2.shape) == (1, 32) and p.size(0) > 0 and not _os.environ.get('GML_NO_HEAD_BIG'))


def head_l1_supported(p, w1, w2):
    R, nin, nh = int(p.size(0)), int(p.size(1)), int(w1.size(0))
    return (p.is_cuda and p.dtype == torch.float32 and R <= 256 and nin <= 64 and nh <= 64 and int(w2.size(0)) == 1
            and nin % 4 == 0 and nh % 4 == 0 and 4 * (R * nin + 2 * R * nh + 2 * R + nh * nin + 256) <= 160 * 1024)


def tall_linear(x, lin):
    return TallLinearFunction.apply(x, lin.weight, lin.bias)


def _ptr32(ptr):
    """segment pointers as the kernels read them: int32, contiguous (a PyG-style int64 ``ptr`` is converted)."""
    if ptr.dtype != torch.int32:
        if ptr.dtype not in (torch.int64, torch.int16, torch.uint8):
            raise TypeError('ptr must be an integer tensor, got %s' % ptr.dtype)
        ptr = ptr.to(torch.int32)
    return ptr.contiguous()


def segment_bcast(g, ptr, nrows, mean=False):
    """gradient of segment_sum: row r of segment s receives g[s] (divided by the segment length for the mean)."""
    ptr = _ptr32(ptr)
    B, F = int(ptr.numel() - 1), int(g.size(1))
    out = torch.empty(nrows, F, dtype=torch.float32, device=g.device)
    with _Timed('pool'):
        _lib.call('gml_segment_bcast', _ptr(g), int(g.stride(0)), _ptr(ptr), _ptr(out), F, B, F, 1 if mean else 0,
                  _stream(g.device))
    return out


_SKIP_LAST_MASK = {}


def skip_last_mask(g):
    """[B, 1] ones with a zero in the last row (cached per (B, device)): the pooled gradient of a batch whose last segment is
    the padding graph of a static-shape batch (GML_POOL_SKIP_LAST) -- the forward writes that row as zeros WITHOUT reading the
    padding nodes, so nothing may flow back to them whatever the head or the loss does with the row."""
    key = (int(g.size(0)), g.device)
    m = _SKIP_LAST_MASK.get(key)
    if m is None:
        m = torch.ones(g.size(0), 1, dtype=g.dtype, device=g.device)
        m[-1] = 0
        _SKIP_LAST_MASK[key] = m
    return m


def segment_sum(x, ptr, mean=False):
    """global_add_pool / global_mean_pool over a batch whose nodes are grouped per graph (ptr [B+1] int32).  mean: bool, or the
    flag word of gml_segment_sum (bit 0 mean, bit 1 = _lib.GML_POOL_SKIP_LAST: the last segment is the padding graph of a
    static-shape batch, its row is written as zeros)."""
    ptr = _ptr32(ptr)
    B, F = int(ptr.numel() - 1), int(x.size(1))
    out = torch.empty(B, F, dtype=torch.float32, device=x.device)
    with _Timed('pool'):
        _lib.call('gml_segment_sum', _ptr(x), int(x.stride(0)), _ptr(ptr), _ptr(out), F, B, F, int(mean),
                  _stream(x.device))
    return out


def segment_sum_mask(x, ptr, mean=False):
    """segment_sum for 32-column rows + the relu pattern of every row (uint32 per row as int32 [N]: bit c = x[row][c] > 0)."""
    ptr = _ptr32(ptr)
    B, F = int(ptr.numel() - 1), int(x.size(1))
    out = torch.empty(B, F, dtype=torch.float32, device=x.device)
    mask = torch.empty(int(x.size(0)), dtype=torch.int32, device=x.device)
    with _Timed('pool'):
        _lib.call('gml_segment_sum_mask', _ptr(x), int(x.stride(0)), _ptr(ptr), _ptr(out), F, _ptr(mask), B, F, int(mean), _stream(x.device))
    return out, mask


def segment_bcast_mask(g, seg, mask, nrelu):
    """the pool's gradient per row with the relu pattern applied: out[row, c] = g[seg[row], c] * (c >= nrelu or bit c of mask[row])."""
    N = int(seg.numel())
    out = torch.empty(N, 32, dtype=torch.float32, device=g.device)
    with _Timed('pool'):
        _lib.call('gml_segment_bcast_mask', _ptr(g), int(g.stride(0)), _ptr(seg), _ptr(mask), _ptr(out), 32, N, 32, int(nrelu), _stream(g.device))
    return out


def segment_max(x, ptr):
    """global_max_pool over a batch whose nodes are grouped per graph: (max [B, F], argmax rows [B, F] int32)."""
    ptr = _ptr32(ptr)
    B, F = int(ptr.numel() - 1), int(x.size(1))
    out = torch.empty(B, F, dtype=torch.float32, device=x.device)
    arg = torch.empty(B, F, dtype=torch.int32, device=x.device)
    with _Timed('pool'):
        _lib.call('gml_segment_max', _ptr(x), int(x.stride(0)), _ptr(ptr), _ptr(out), F, _ptr(arg), B, F, _stream(x.device))
    return out, arg


def segment_max_bwd(g, ptr, arg, nrows):
    ptr = _ptr32(ptr)
    B, F = int(ptr.numel() - 1), int(g.size(1))
    out = torch.empty(nrows, F, dtype=torch.float32, device=g.device)
    with _Timed('pool'):
        _lib.call('gml_segment_max_bwd', _ptr(g), int(g.stride(0)), _ptr(ptr), _ptr(arg), _ptr(out), F, B, F, _stream(g.device))
    return out


def _bwd_plan(csr, S, Fin, Fout):
    """(flags, ginfo, (max_edges, max_window), workspace bytes) of the fused backward for this shape, or None."""
    L = _lib.lib()
    for flags in ((_lib.GML_F32_MFMA,) if exact_mode() else (0, _lib.GML_F32_MFMA)):
        rows = int(L.gml_spectconv_bwd_group_rows(int(S), int(Fin), int(Fout), flags))
        if rows == 0:
            continue
        if rows == 128:
            ginfo, gmax = csr.ginfo_t128, csr.gmax_t128
        elif rows == _lib.GML_GROUPS64_RANKED:
            ginfo, gmax = csr.ranked64_t()
        else:
            ginfo, gmax = csr.ginfo_t, csr.gmax_t
        nbytes = int(L.gml_spectconv_bwd_workspace_bytes(csr.N, int(S), int(Fin), int(Fout), gmax[0], gmax[1], flags))
        if nbytes > 0:
            return flags, ginfo, gmax, nbytes, rows
    return None


def fused_bwd_available(csr, S, Fin, Fout):
    return _bwd_plan(csr, S, Fin, Fout) is not None


def conv_bwd_takes_dz(csr, S, Fin, Fout, nmix):
    """True when the fused backward of this shape can form  dx = conv part + dz wmix  itself (gml_spectconv_bwd_mix)."""
    if _os.environ.get('GML_NO_DZ') or not (1 <= nmix <= 4):
        return False
    plan = _bwd_plan(csr, S, Fin, Fout)
    return plan is not None and bool(_lib.lib().gml_spectconv_bwd_mix_supported(int(S), int(Fin), int(Fout), int(nmix), plan[0]))


BWD_DMA = _os.environ.get('GML_BWD_DMA') == '1'     # fused backward on the LDS-DMA ring kernel (bwd4) where it applies; A/B switch
BWD_HAD = _os.environ.get('GML_BWD_HAD', '1') != '0'  # ML3Layer output stage inside the conv backward (gml_spectconv_bwd_had); A/B switch


def conv_bwd_had_parts(csr, S, Fin, Fout, F2, want_dx=True):
    """Partial rows of gml_spectconv_bwd_had for this shape (0: the shape / the build has no fused output stage)."""
    if not BWD_HAD or BWD_DMA or exact_mode():
        return 0
    plan = _bwd_plan(csr, S, Fin, Fout)
    if plan is None or plan[4] != 128:
        return 0
    return int(_lib.lib().gml_spectconv_bwd_had_parts(csr.N, int(S), int(Fin), int(Fout), int(F2), 1 if want_dx else 0, plan[2][0], plan[2][1],
                                                      plan[0]))


def fused_conv_bwd(csr, val_t, x, G, weight, need_x, need_val, need_w, dx_accum_into=None, mix=None, relu_cols=0, had=None):
    """one launch: dX, dval (source order), dW.  val_t: supports in source order.  mix = (dz [N, 4], wmix [nmix, Fin]):
    dX = conv part + dz wmix (instead of accumulating into a dx another kernel wrote).  relu_cols (with mix): dX[:, f] for
    f < relu_cols is written multiplied by (x[:, f] > 0) (gml_spectconv_bwd_mix_relu: the relu of the ML3Layer below).
    had = {'w': (w11, b11, w12, b12), 'parts': rows}: the ML3Layer output stage inside the launch (gml_spectconv_bwd_had; G is then a
    view of the pre-masked [N, >= Fout + 2] gradient whose columns Fout, Fout + 1 are the Hadamard units'); had['out'] receives
    (dcb, dw11, db11, dw12, db12)."""
    S, Fin, Fout = weight.shape
    dev = x.device
    flags, ginfo, gmax, nbytes, _ = _bwd_plan(csr, S, Fin, Fout)
    if BWD_DMA:
        flags |= _lib.GML_DMA_RING
    ld4 = (Fout + 3) // 4 * 4
    if G.stride(0) % 4 != 0 or G.stride(0) < ld4 or G.data_ptr() % 16 != 0:
        # the 8-wave kernel reads the g window as float4: rows padded (with zeros) to a multiple of 4 floats
        Gp = torch.zeros(csr.N, ld4, dtype=torch.float32, device=dev)
        Gp[:, :Fout] = G
        G = Gp[:, :Fout]
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev) if need_w else None
    if need_x and dx_accum_into is not None:                 # dx already holds another branch's contribution
        dx, flags = dx_accum_into, flags | _lib.GML_ACCUM
    else:
        dx = torch.empty(csr.N, Fin, dtype=torch.float32, device=dev) if need_x else None
    dval_t = torch.empty(csr.E, S, dtype=torch.float32, device=dev) if need_val else None
    dw = torch.empty(S, Fin, Fout, dtype=torch.float32, device=dev) if need_w else None
    q, f = conv_cost_bwd(csr.N, csr.E, S, Fin, Fout, need_x, need_val) if PROFILE is not None else (0, 0)
    fq = _fold_queue() if need_w else None
    if fq is not None:                                       # the dW partials stay in ws; one launch folds every layer's at scope exit
        flags |= _lib.GML_NO_FOLD
        flat = torch.empty(S * Fin * Fout, dtype=torch.float32, device=dev)       # (a view of it goes to autograd: see edge_mlp_bwd)
        fq.append((ws, nbytes // (4 * S * Fin * Fout), S * Fin * Fout, [(flat, flat.numel())]))
    if had is not None:
        w11, b11, w12, b12 = had['w']
        npart, parts = 4 * Fin + 4 + Fout, int(had['parts'])
        hws = torch.empty(parts * npart, dtype=torch.float32, device=dev)
        hfq = _fold_queue()
        if hfq is not None:                                  # (same layout as gml_ml3_split_bwd's partial rows: same fold job)
            hflat = torch.empty(npart, dtype=torch.float32, device=dev)
            hfq.append((hws, parts, npart, [(hflat, npart)]))
            dw11, dw12 = hflat[:2 * Fin].view_as(w11), hflat[2 * Fin:4 * Fin].view_as(w12)
            db11 = hflat[4 * Fin:4 * Fin + 2] if b11 is not None else None
            db12 = hflat[4 * Fin + 2:4 * Fin + 4] if b12 is not None else None
            dcb = hflat[4 * Fin + 4:]
            kd = (None,) * 5
        else:
            dw11, dw12 = torch.empty_like(w11), torch.empty_like(w12)
            db11 = torch.empty_like(b11) if b11 is not None else None
            db12 = torch.empty_like(b12) if b12 is not None else None
            dcb = torch.empty(Fout, dtype=torch.float32, device=dev)
            kd = (dcb, dw11, db11, dw12, db12)
        had['out'] = (dcb, dw11, db11, dw12, db12)
        with _Timed('spectconv_bwd', q + 8 * csr.N, f):        # (+ the own rows' two Hadamard columns; the rest it reads anyway)
            _lib.call('gml_spectconv_bwd_had', _ptr(csr.rowptr_t), _ptr(csr.col_t), _ptr(ginfo), _ptr(val_t),
                      _ptr(x), int(x.stride(0)), _ptr(G), int(G.stride(0)), _ptr(weight), _ptr(dx), Fin, _ptr(dval_t),
                      _ptr(dw), _ptr(w11), _ptr(b11), _ptr(w12), _ptr(b12), int(relu_cols), _ptr(kd[0]), _ptr(kd[1]), _ptr(kd[2]),
                      _ptr(kd[3]), _ptr(kd[4]), csr.N, S, Fin, Fout, 2, gmax[0], gmax[1], flags, _ptr(ws),
                      ws.numel() if ws is not None else 0, _ptr(hws), hws.numel() * 4, _stream(dev))
        if fq is not None:
            dw = flat.view(S, Fin, Fout)
        return dx, dval_t, dw
    with _Timed('spectconv_bwd', q, f):
        if mix is not None:
            dz, wmix = mix
            if isinstance(wmix, tuple):                      # (fc11.weight, fc12.weight) as they are stored: no concatenation launch
                wa, wb = wmix
                _lib.call('gml_spectconv_bwd_mix_relu2', _ptr(csr.rowptr_t), _ptr(csr.col_t), _ptr(ginfo), _ptr(val_t),
                          _ptr(x), int(x.stride(0)), _ptr(G), int(G.stride(0)), _ptr(weight), _ptr(dx), Fin, _ptr(dval_t),
                          _ptr(dw), _ptr(dz), _ptr(wa), int(wa.size(0)), _ptr(wb), int(wb.size(0)), int(relu_cols), csr.N, S, Fin, Fout,
                          gmax[0], gmax[1], flags, _ptr(ws), ws.numel() if ws is not None else 0, _stream(dev))
            else:
                _lib.call('gml_spectconv_bwd_mix_relu', _ptr(csr.rowptr_t), _ptr(csr.col_t), _ptr(ginfo), _ptr(val_t),
                          _ptr(x), int(x.stride(0)), _ptr(G), int(G.stride(0)), _ptr(weight), _ptr(dx), Fin, _ptr(dval_t),
                          _ptr(dw), _ptr(dz), _ptr(wmix), int(wmix.size(0)), int(relu_cols), csr.N, S, Fin, Fout, gmax[0], gmax[1], flags,
                          _ptr(ws), ws.numel() if ws is not None else 0, _stream(dev))
        else:
            assert relu_cols == 0, 'the relu hand-over rides on the dz form of the backward'
            _lib.call('gml_spectconv_bwd', _ptr(csr.rowptr_t), _ptr(csr.col_t), _ptr(ginfo), _ptr(val_t),
                      _ptr(x), int(x.stride(0)), _ptr(G), int(G.stride(0)), _ptr(weight), _ptr(dx), Fin, _ptr(dval_t),
                      _ptr(dw), csr.N, S, Fin, Fout, gmax[0], gmax[1], flags, _ptr(ws),
                      ws.numel() if ws is not None else 0, _stream(dev))
    if fq is not None:
        dw = flat.view(S, Fin, Fout)
    return dx, dval_t, dw


def split48_plan(csr, S, Fin, Fout):
    """48-wide layers (sr25.py:252-262, mutag.py:272-288: hidden width 32 + 16 / 24 + 24) on the 8-wave bf16x3 backward, which is
    compiled for Fin <= 32: two launches over the feature slices [0, 32) and [32, Fin).  dX and dW split by input feature; dval is
    linear in x, so the second launch adds its share (GML_DVAL_ACCUM).  The edge loop runs twice -- still ~2 x faster than the
    64-row f32-MFMA kernel these layers ran on (tools/bench_configs.py).  Returns the two plans or None."""
    if exact_mode() or _os.environ.get('GML_NO_SPLIT48') or not (32 < Fin <= 48) or (Fin - 32) % 4 != 0 or (S == 8 and Fout > 16):   # (S = 8, 32 columns: no dval += in that kernel)
        return None
    pa, pb = _bwd_plan(csr, S, 32, Fout), _bwd_plan(csr, S, Fin - 32, Fout)
    if pa is None or pb is None or pa[4] != 128 or pb[4] != 128 or (pa[0] | pb[0]) & _lib.GML_F32_MFMA:
        return None
    return pa, pb


def fused_conv_bwd_split48(csr, val_t, x, G, weight, need_x, need_val, need_w, dx_accum_into, plans):
    S, Fin, Fout = weight.shape
    dev = x.device
    ld4 = (Fout + 3) // 4 * 4
    if G.stride(0) % 4 != 0 or G.stride(0) < ld4 or G.data_ptr() % 16 != 0:
        Gp = torch.zeros(csr.N, ld4, dtype=torch.float32, device=dev)
        Gp[:, :Fout] = G
        G = Gp[:, :Fout]
    dx = dx_accum_into if (need_x and dx_accum_into is not None) else (torch.empty(csr.N, Fin, dtype=torch.float32, device=dev) if need_x else None)
    dval_t = torch.empty(csr.E, S, dtype=torch.float32, device=dev) if need_val else None
    dws = []
    q, f = conv_cost_bwd(csr.N, csr.E, S, Fin, Fout, need_x, need_val) if PROFILE is not None else (0, 0)
    with _Timed('spectconv_bwd', q, f):
        for part, (f0, f1) in enumerate(((0, 32), (32, Fin))):
            flags, ginfo, gmax, nbytes, _ = plans[part]
            if need_x and dx_accum_into is not None:
                flags |= _lib.GML_ACCUM
            if part == 1 and need_val:
                flags |= _lib.GML_DVAL_ACCUM
            w_p = weight[:, f0:f1, :].contiguous()
            dw_p = torch.empty(S, f1 - f0, Fout, dtype=torch.float32, device=dev) if need_w else None
            ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev) if need_w else None
            _lib.call('gml_spectconv_bwd', _ptr(csr.rowptr_t), _ptr(csr.col_t), _ptr(ginfo), _ptr(val_t),
                      _off(x, f0), int(x.stride(0)), _ptr(G), int(G.stride(0)), _ptr(w_p), _off(dx, f0) if dx is not None else _ptr(None), Fin,
                      _ptr(dval_t), _ptr(dw_p), csr.N, S, f1 - f0, Fout, gmax[0], gmax[1], flags, _ptr(ws),
                      ws.numel() if ws is not None else 0, _stream(dev))
            dws.append(dw_p)
    dw = torch.cat(dws, 1) if need_w else None
    return dx, dval_t, dw


# ---------------------------------------------------------------------------- shared backward pieces
def _conv_backward(csr, x, val, weight, G, need_x, need_val, need_w, val_t=None, want_source_order=False,
                   dx_accum_into=None, mix=None, relu_cols=0, had=None):
    """G [N,Fout] contiguous = gradient at the (pre-activation) conv output.
    Returns dx, dval, dw; dval is in source order when want_source_order (and the fused kernel ran).
    dx_accum_into: [N,Fin] buffer that already holds a partial dx; the conv contribution is added to it."""
    S, Fin, Fout = weight.shape
    N = csr.N
    dx = dval = dw = None
    if G.stride(1) != 1:
        G = G.contiguous()
    # 33 .. 48 input features: ONE launch of the 8-wave kernel where the library has that shape (round 5: S = 4, 6; GML_BWD_WIDE48=0
    # turns it off), else two launches over the feature slices
    one48 = None
    if 32 < Fin <= 48 and not exact_mode():
        one48 = _bwd_plan(csr, S, Fin, Fout)
        if one48 is not None and (one48[4] != 128 or (one48[0] & _lib.GML_F32_MFMA)):
            one48 = None
    sp48 = split48_plan(csr, S, Fin, Fout) if (one48 is None and mix is None and relu_cols == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0) else None
    if sp48 is not None or fused_bwd_available(csr, S, Fin, Fout):
        if val_t is None:
            with _Timed('val_to_source_order'):
                val_t = csr.to_source_order(val)
        if sp48 is not None:
            _path('conv_bwd', 'fused (group kind 128), two launches over the feature slices [0, 32) and [32, %d)' % Fin, S, Fin, Fout)
            dx, dval_t, dw = fused_conv_bwd_split48(csr, val_t, x, G, weight, need_x, need_val, need_w, dx_accum_into, sp48)
        else:
            _path('conv_bwd', 'fused (group kind %d)' % _bwd_plan(csr, S, Fin, Fout)[4], S, Fin, Fout)
            dx, dval_t, dw = fused_conv_bwd(csr, val_t, x, G, weight, need_x, need_val, need_w, dx_accum_into, mix, relu_cols, had)
        if need_val and not want_source_order:
            with _Timed('dval_from_source_order'):
                dval_t = csr.from_source_order(dval_t)
        return dx, dval_t, dw, (want_source_order and need_val)
    _path('conv_bwd', 'UNFUSED composition (transposed forward + SpMM + SDDMM + GEMMs)', S, Fin, Fout)
    if not G.is_contiguous():
        G = G.contiguous()
    if need_x:
        dx = dx_accum_into if dx_accum_into is not None else torch.empty(N, Fin, dtype=torch.float32, device=x.device)
        # dX = sum_s A_s (G W_s^T): rows keyed by SOURCE, features = G, weight element (s, o, f) = W[s, f, o]
        if val_t is None:
            with _Timed('val_to_source_order'):
                val_t = csr.to_source_order(val)
        fused_conv(csr.rowptr_t, csr.col_t, csr.ginfo_t, None, val_t, G, Fout, weight, (Fin * Fout, 1, Fout), None,
                   dx, Fin, N, S, Fout, Fin, _lib.GML_ACCUM if dx_accum_into is not None else 0, tag='spectconv_dx')
    if need_w:
        with _Timed('dw_spmm'):
            h = spmm(csr, val, x, S, Fin)                                # [N, S*Fin]
        with _Timed('dw_gemm'):
            dw = torch.mm(h.t(), G).view(S, Fin, Fout)
    if need_val:
        with _Timed('dval_gemm'):
            gw = torch.mm(G, weight.view(S * Fin, Fout).t())             # [N, S*Fin]
        with _Timed('dval_sddmm'):
            dval = sddmm(csr, x, gw, S, Fin)
    return dx, dval, dw, False


class SpectConvFunction(torch.autograd.Function):
    """out = act( sum_s (A_s^T x) W_s + bias ),  val = supports in target-sorted order."""

    @staticmethod
    def forward(ctx, x, val, weight, bias, csr, relu):
        x, val, weight = _f32rows(x, 'x'), _f32c(val, 'edge_attr'), _f32c(weight, 'weight')
        S, Fin, Fout = weight.shape
        if x.size(1) != Fin or val.size(1) < S or val.size(0) != csr.E or x.size(0) != csr.N:
            raise ValueError('shape mismatch: x %s, edge_attr %s, weight %s, graph N=%d E=%d'
                             % (tuple(x.shape), tuple(val.shape), tuple(weight.shape), csr.N, csr.E))
        if val.size(1) != S:
            val = val[:, :S].contiguous()
        if bias is not None:
            bias = _f32c(bias, 'bias')
        with torch.cuda.device(x.device):
            out = torch.empty(csr.N, Fout, dtype=torch.float32, device=x.device)
            x = rows4(x)
            gi, gflag = fwd_groups(csr, x, S, Fin, Fout)
            fused_conv(csr.rowptr, csr.col, gi, None, val, x, int(x.stride(0)), weight, (Fin * Fout, Fout, 1), bias, out,
                       Fout, csr.N, S, Fin, Fout, (_lib.GML_RELU if relu else 0) | gflag)
        ctx.csr, ctx.relu, ctx.has_bias = csr, relu, bias is not None
        ctx.exact = bool(exact_mode())
        ctx.save_for_backward(x, val, weight, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        with exact_products(_bwd_exact(ctx)):
            return SpectConvFunction._backward(ctx, gout)

    @staticmethod
    def _backward(ctx, gout):
        x, val, weight, out = ctx.saved_tensors
        csr = ctx.csr
        gout = _f32c(gout, 'grad_out')
        with torch.cuda.device(x.device):
            Fout = weight.size(2)
            need = ctx.needs_input_grad
            want_db = ctx.has_bias and need[3]
            db = None
            r = ml3_split_bwd(gout, out, Fout, need_dcb=want_db) if ctx.relu else None
            if r is not None:
                G, db = r[0], r[2]
            else:
                G = relu_bwd(gout, 0, Fout, out, Fout, csr.N, Fout) if ctx.relu else gout
            dx, dval, dw, _ = _conv_backward(csr, x, val, weight, G, need[0], need[1], need[2])
            if want_db and db is None:
                db = G.sum(0)
        return dx, dval, dw, db, None, None


class ChainToken(object):
    """Hand-over between two stacked ML3Layers (Zinc12k.py:338-341: x = conv2(conv1(x, ...), ...)): the upper layer's conv
    backward holds the rows of x = [relu(conv) | Hadamard columns] of the lower layer anyway, so it writes dx already
    multiplied by that relu's mask (`premasked`), and the lower layer's output-stage backward then neither reads its saved
    output nor writes a second [N, C] array.  Valid only when the upper layer is the ONLY consumer of that tensor --
    ML3Layer.chain_after() is how a model says so.  cols: relu columns of the lower layer's output (nout1)."""
    __slots__ = ('cols', 'premasked')

    def __init__(self, cols):
        self.cols, self.premasked = int(cols), False


CHAIN = not _os.environ.get('GML_NO_CHAIN')      # A/B switch for the relu hand-over between stacked ML3Layers


class ML3LayerFunction(torch.autograd.Function):
    """Whole ML3Layer.forward (libs/spect_conv.py:204-212) with the concat written in place:
         ea' = edge-MLP(val)                      (learnedge)
         out[:, :nout1] = relu(SpectConv(x, ea')) (fused kernel, leading dimension nout1+nout2)
         out[:, nout1:] = tanh(fc11 x) * tanh(fc12 x)
    """

    @staticmethod
    def forward(ctx, x, val, w1, w2, w3, w4, cw, cb, w11, b11, w12, b12, csr, learnedge, nout2, val_is_source=False,
                pool_ptr=None, pool_seg=None, pool_mean=False, chain_in=None, chain_out=None, ea_pre=None, stack=None):
        # ea_pre: this layer's edge-branch output in source order, already computed by the first layer of its stack; stack =
        # (weights of the layers stacked on this one [(w1, w2, w3, w4)], list that receives their edge-branch outputs): the
        # layers all read the same raw supports, one pass serves them (edge_mlp_fwd_stack)
        # chain_in / chain_out (ChainToken): x is the output of the ML3Layer that owns chain_in and has no other consumer /
        # this layer's own token, which the consumer of its output sets (see ChainToken)
        # pool_ptr / pool_seg (int32 [B+1] / [N]): the layer is directly followed by global_add_pool / global_mean_pool
        # (Zinc12k.py:343): the pooled [B, C] tensor is returned and the pool's gradient is never expanded to [N, C]
        # val_is_source: val holds the raw supports in SOURCE order (the caller checked ml3_edge_in_source_order and that they
        # carry no gradient); otherwise target-sorted order
        x, val, cw = _f32rows(x, 'x'), _f32c(val, 'edge_attr'), _f32c(cw, 'conv1.weight')
        S, Fin, nout1 = cw.shape
        N = csr.N
        if x.size(0) != N or x.size(1) != Fin or val.size(0) != csr.E:
            raise ValueError('shape mismatch: x %s, edge_attr %s, conv1.weight %s, graph N=%d E=%d'
                             % (tuple(x.shape), tuple(val.shape), tuple(cw.shape), N, csr.E))
        C = nout1 + nout2
        with torch.cuda.device(x.device):
            x = rows4(x)
            epos = None
            if learnedge:
                w1, w2, w3, w4 = (_f32c(w1, 'fc1_1.weight'), _f32c(w2, 'fc1_2.weight'), _f32c(w3, 'fc1_3.weight'),
                                  _f32c(w4, 'fc1_4.weight'))
                # When the fused backward will run, the edge branch lives in SOURCE order (the order that kernel walks): the raw
                # supports in that order are per-batch data (for source-sorted input: the input itself), the branch writes
                # its output once, and the forward conv gathers the rows through tpos (target position -> source position).
                # (r02: emitting both orders from the edge kernel cost it 37 % -- it is HBM-bound and the second copy was a
                # scattered 32-byte-row write.)
                fused_b = any(ctx.needs_input_grad) and fused_bwd_available(csr, S, Fin, nout1)
                src_order = bool(val_is_source) or (fused_b and fwd_gathers(S, Fin, nout1))
                # (forward kernels without the gather: both orders from the edge kernel, second copy scattered through tpos)
                dual = fused_b and not src_order and val.numel() * 4 < 0xffffff00
                _path('edge', 'exact family (one edge per lane, fp32)' if exact_mode() else 'matrix-core chain' if max(val.size(1), w4.size(0)) <= 8 else ('matrix-core chain16' if max(val.size(1), w4.size(0)) <= 16 and not EDGE_VALU else 'VALU kernels'), val.size(1), '-', w4.size(0))
                if src_order:
                    val_s = val if val_is_source else csr.to_source_order(val, cache=not val.requires_grad)
                    ea_t = ea_pre if val_is_source else None
                    if ea_t is None and stack is not None and val_is_source:
                        ws_ = [(w1, w2, w3, w4)] + [tuple(_f32c(t, 'edge branch weight') for t in w) for w in stack[0]]
                        with _Timed('edge_mlp_fwd', 4 * val.numel() * (1 + len(ws_)), 20 * val.size(0) * val.size(1) ** 2 * len(ws_)):
                            outs = edge_mlp_fwd_stack(val_s, csr.presplit(val_s), ws_, csr.sym_index(val_s) if EDGE_SYM else None)
                        if outs is not None:
                            _path('edge', 'stack of %d layers in one pass' % len(ws_), val.size(1), '-', w4.size(0))
                            ea_t = outs[0]
                            stack[1].extend(outs[1:])
                    if ea_t is None:
                        with _Timed('edge_mlp_fwd', 4 * val.numel() * 2, 20 * val.size(0) * val.size(1) ** 2):
                            sym_ = csr.sym_index(val_s) if (EDGE_SYM and not val.requires_grad) else None
                            one = edge_mlp_fwd_stack(val_s, None, [(w1, w2, w3, w4)], sym_) if sym_ is not None else None
                            ea_t = one[0] if one is not None else edge_mlp_fwd(val_s, w1, w2, w3, w4, None, csr.presplit(val_s))[0]
                    ea, epos = ea_t, csr.tpos
                else:
                    with _Timed('edge_mlp_fwd', 4 * val.numel() * (3 if dual else 2), 20 * val.size(0) * val.size(1) ** 2):
                        # (inference / unfused training: target order; unique support rows only where the batch pairs up)
                        sym_ = csr.sym_index(val, 'target') if (EDGE_SYM and not dual and not val.requires_grad and w4.size(0) == val.size(1)) else None
                        one = edge_mlp_fwd_stack(val, None, [(w1, w2, w3, w4)], sym_) if sym_ is not None else None
                        if one is not None:
                            ea, ea_t = one[0], None
                        else:
                            ea, ea_t = edge_mlp_fwd(val, w1, w2, w3, w4, csr.tpos if dual else None, csr.presplit(val))
            else:
                ea, ea_t = val, None
            if ea.size(1) != S:
                raise ValueError('conv1 expects %d supports, edge branch produced %d' % (S, ea.size(1)))
            out = torch.empty(N, C, dtype=torch.float32, device=x.device)
            cb_ = _f32c(cb, 'conv1.bias') if cb is not None else None
            gi, gflag = fwd_groups(csr, x, S, Fin, nout1)
            if nout2 > 0:
                w11, b11, w12, b12 = (_f32c(w11, 'fc11.weight'), _f32c(b11, 'fc11.bias'), _f32c(w12, 'fc12.weight'),
                                      _f32c(b12, 'fc12.bias'))
            q, f = conv_cost(N, int(ea.size(0)), S, Fin, nout1) if PROFILE is not None else (0, 0)
            mixk = nout2 > 0 and node_mix_native(Fin, nout2)
            if nout2 > 0:
                _path('hadamard', 'fused kernels' if mixk else 'library GEMMs (ninp > 64 or nout2 > 24)', '-', Fin, nout2)
            with _Timed('spectconv_fwd', q, f):                    # conv (+ Hadamard branch of the same rows)
                _lib.call('gml_ml3_fwd', _ptr(csr.rowptr), _ptr(csr.col), _ptr(gi), _ptr(epos), _ptr(ea), _ptr(x), int(x.stride(0)), _ptr(cw),
                          Fin * nout1, nout1, 1, _ptr(cb_), _ptr(w11 if mixk else None),
                          _ptr(b11 if mixk else None), _ptr(w12 if mixk else None),
                          _ptr(b12 if mixk else None), _ptr(out), C, N, S, Fin, nout1, int(nout2) if mixk else 0,
                          _lib.GML_RELU | gflag | _fwd_arith_flags(), _stream(x.device))
            if nout2 > 0 and not mixk:
                # ninp > 64 or nout2 > 24 (ptc.py:331-338 has ninp = 80): two plain library GEMMs + elementwise
                out[:, nout1:] = torch.tanh(_linear(x, w11, b11)) * torch.tanh(_linear(x, w12, b12))
        ctx.csr, ctx.learnedge, ctx.nout2, ctx.has_cb = csr, learnedge, nout2, cb is not None
        ctx.exact = bool(exact_mode())
        ctx.src_order = epos is not None
        ctx.val_is_source = bool(val_is_source)
        ctx.pool = (pool_ptr, pool_seg, int(pool_mean)) if pool_ptr is not None else None     # bit 0 mean, bit 1 GML_POOL_SKIP_LAST
        ctx.chain_in = chain_in if (CHAIN and chain_in is not None and chain_in.cols <= Fin) else None
        ctx.chain_out = chain_out if (CHAIN and ctx.pool is None) else None
        ctx.save_for_backward(x, val, (ea if learnedge and epos is None else None), w1, w2, w3, w4, cw, w11, b11, w12, b12, out, ea_t)
        ctx.pool_mask = None
        if ctx.pool is not None:
            # ZINC's 30 + 2 top layer: the pool also leaves the relu pattern of every row (4 bytes) -- the backward then takes the
            # pre-masked road with the output stage inside the conv backward instead of reading the saved output
            if (BWD_HAD and mixk and nout1 == 30 and nout2 == 2 and out.stride(0) == 32 and not exact_mode()
                    and any(ctx.needs_input_grad)):              # (inference: the plain pool)
                pooled, ctx.pool_mask = segment_sum_mask(out, pool_ptr, int(pool_mean) & 3)
            else:
                pooled = segment_sum(out, pool_ptr, int(pool_mean) & 3)
            if int(pool_mean) & 2:
                skip_last_mask(pooled)                         # (created outside any later graph capture of the backward)
            return pooled
        return out

    @staticmethod
    def backward(ctx, gy):
        with exact_products(_bwd_exact(ctx)):
            return ML3LayerFunction._backward(ctx, gy)

    @staticmethod
    def _backward(ctx, gy):
        x, val, ea, w1, w2, w3, w4, cw, w11, b11, w12, b12, out, ea_t = ctx.saved_tensors
        csr, learnedge, nout2 = ctx.csr, ctx.learnedge, ctx.nout2
        S, Fin, nout1 = cw.shape
        N, C = csr.N, nout1 + nout2
        need = ctx.needs_input_grad
        gy = _f32c(gy, 'grad_out')
        if not learnedge:
            ea = val
        g = [None] * 23
        gy_seg = None
        pre_top = False
        if ctx.pool is not None:                               # gy is the POOLED gradient [B, C]
            pptr, pseg, pmean = ctx.pool
            # GML_POOL_SKIP_LAST: the padding graph's pooled row was written, not computed -- its gradient must not reach the nodes.
            # Bit 2 (4): the caller vouches that the loss gives that row a ZERO gradient already (models.zinc_step_loss: the fused
            # head + loss ignore rows beyond the labelled ones) -- the masking launch is skipped
            if (pmean & 2) and not (pmean & 4):
                gy = gy * skip_last_mask(gy)
            pmean &= 1
            mixk_ = nout2 > 0 and node_mix_native(Fin, nout2)
            nb_ = int(_lib.lib().gml_ml3_split_bwd_workspace_bytes(int(N), Fin if mixk_ else 0, int(nout1), int(nout2) if mixk_ else 0))
            if nb_ > 0 and (mixk_ or nout2 == 0) and not _os.environ.get('GML_NO_POOL_FUSE'):
                if pmean:
                    cnt = (pptr[1:] - pptr[:-1]).clamp(min=1).to(gy.dtype).unsqueeze(1)
                    gy = gy / cnt
                gy_seg = pseg
                if (ctx.pool_mask is not None and BWD_HAD and mixk_ and nout2 == 2 and need[0] and need[6] and need[8] and need[10]
                        and gy.stride(1) == 1 and gy.stride(0) % 4 == 0 and gy.data_ptr() % 16 == 0
                        and conv_bwd_takes_dz(csr, S, Fin, nout1, 4) and conv_bwd_had_parts(csr, S, Fin, nout1, nout2) > 0):
                    # the pool's gradient per row with the relu pattern applied (4 + 4 bytes read, 128 written per row): the layer
                    # then runs like the ones below it -- pre-masked, output stage inside the conv backward
                    gy = segment_bcast_mask(gy, pseg, ctx.pool_mask, nout1)
                    gy_seg, pre_top = None, True
            else:
                gy = segment_bcast(gy, pptr, N, pmean)
        with torch.cuda.device(x.device):
            need_val = need[1] or (learnedge and any(need[2:6]))
            if not learnedge and fused_bwd_available(csr, S, Fin, nout1):
                ea_t = csr.to_source_order(val, cache=not val.requires_grad)   # raw supports: per-batch data (trained ones change)
            want_cb = ctx.has_cb and need[7]
            mixk = nout2 > 0 and node_mix_native(Fin, nout2)
            # 2 nout2 <= 4 (Zinc12k.py's 30+2 layers): the Hadamard branch hands its share of dx to the conv backward as 4 numbers
            # per row (dz) instead of writing a [N, Fin] array the conv kernel reads back
            use_dz = bool(mixk and need[0] and conv_bwd_takes_dz(csr, S, Fin, nout1, 2 * nout2))
            # the consumer of this layer's output already applied the relu mask to gy (ChainToken); read once, then cleared
            pre = (ctx.chain_out is not None and ctx.chain_out.premasked and gy_seg is None) or pre_top
            if ctx.chain_out is not None:
                ctx.chain_out.premasked = False
            # ZINC's 30 + 2 layers with a pre-masked gradient: the output stage runs INSIDE the conv backward (one launch, no dz array,
            # no second pass over gy and x)
            hparts = 0
            x4 = x.stride(0) % 4 == 0 and x.stride(0) >= (Fin + 3) // 4 * 4 and x.data_ptr() % 16 == 0     # float4-readable x rows
            if ((use_dz or (mixk and not need[0] and x4)) and pre and nout2 == 2 and need[6] and need[8] and need[10] and gy.stride(1) == 1
                    and gy.stride(0) % 4 == 0 and gy.stride(0) >= C and gy.data_ptr() % 16 == 0):
                hparts = conv_bwd_had_parts(csr, S, Fin, nout1, nout2, want_dx=need[0])
            if hparts > 0:
                had = {'w': (w11, b11, w12, b12), 'parts': hparts}
                rc = ctx.chain_in.cols if (need[0] and ctx.chain_in is not None and not BWD_DMA) else 0
                dx, dea, dcw, dea_src = _conv_backward(csr, x, ea, cw, gy[:, :nout1], need[0], need_val, True, val_t=ea_t,
                                                       want_source_order=learnedge, relu_cols=rc, had=had)
                if rc:
                    ctx.chain_in.premasked = True
                g[6] = dcw
                dcb_, g[8], g[9], g[10], g[11] = had['out']
                g[7] = dcb_ if want_cb else None
                r = False                                    # (done: neither branch below)
            else:
                r = ml3_split_bwd(gy, out, nout1, x, w11, b11, w12, b12, need_dx=need[0], need_dcb=want_cb, dz_out=use_dz, gy_seg=gy_seg,
                                  premasked=pre) \
                    if mixk else ml3_split_bwd(gy, out, nout1, need_dcb=want_cb, gy_seg=gy_seg, premasked=pre)
            if r is False:
                pass
            elif r is not None:
                # one pass: relu mask, conv1.bias gradient, Hadamard branch (its dx written, conv adds to it)
                G, dx0, g[7], g[8], g[9], g[10], g[11] = r
                mix = (dx0, (w11.detach().contiguous(), w12.detach().contiguous())) if use_dz else None    # (rows of wmix from both arrays)
                # x is the lower ML3Layer's output and only this layer consumes it: hand its relu mask back inside dx
                rc = ctx.chain_in.cols if (use_dz and ctx.chain_in is not None and not BWD_DMA) else 0
                dx, dea, dcw, dea_src = _conv_backward(csr, x, ea, cw, G, need[0], need_val, need[6], val_t=ea_t,
                                                       want_source_order=learnedge,
                                                       dx_accum_into=None if use_dz else dx0, mix=mix, relu_cols=rc)
                if rc:
                    ctx.chain_in.premasked = True
                g[6] = dcw
            else:
                assert gy_seg is None, 'pooled gradient without the one-pass output-stage kernel'
                G = relu_bwd(gy, 0, C, out, C, N, nout1)
                dx, dea, dcw, dea_src = _conv_backward(csr, x, ea, cw, G, need[0], need_val, need[6], val_t=ea_t,
                                                       want_source_order=learnedge)
                g[6] = dcw
                if want_cb:
                    g[7] = G.sum(0)
                if mixk:
                    nbytes = int(_lib.lib().gml_node_mix_bwd_workspace_bytes(N, Fin, nout2))
                    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
                    g[8], g[9], g[10], g[11] = (torch.empty_like(w11), torch.empty_like(b11), torch.empty_like(w12),
                                                torch.empty_like(b12))
                    with _Timed('node_mix_bwd'):
                        _lib.call('gml_node_mix_bwd', _ptr(x), int(x.stride(0)), _ptr(w11), _ptr(b11), _ptr(w12), _ptr(b12),
                                  _off(gy, nout1), C, _ptr(dx) if need[0] else _ptr(None), Fin, _ptr(g[8]), _ptr(g[9]),
                                  _ptr(g[10]), _ptr(g[11]), N, Fin, nout2, _ptr(ws), ws.numel(), _stream(x.device))
            if nout2 > 0 and not mixk:                              # wide Hadamard branch: library GEMMs (see forward)
                ta, tb = torch.tanh(_linear(x, w11, b11)), torch.tanh(_linear(x, w12, b12))
                g2 = gy[:, nout1:]
                ga, gb = g2 * tb * (1 - ta * ta), g2 * ta * (1 - tb * tb)
                g[8], g[10] = ga.t().mm(x), gb.t().mm(x)
                g[9] = ga.sum(0) if b11 is not None else None
                g[11] = gb.sum(0) if b12 is not None else None
                if need[0]:
                    dx = dx.addmm_(ga, w11).addmm_(gb, w12)
            g[0] = dx
            if learnedge:
                if need_val:
                    # the edge MLP is per-edge, so it can run in whichever order dea arrived in
                    val_in = val if ctx.val_is_source else (csr.to_source_order(val, cache=not val.requires_grad) if dea_src else val)
                    with _Timed('edge_mlp_bwd', 4 * val.numel() * 2, 60 * val.size(0) * val.size(1) ** 2):
                        gin, g[2], g[3], g[4], g[5] = edge_mlp_bwd(val_in, w1, w2, w3, w4, dea, need[1], csr.presplit(val_in),
                                                                   csr.sym_index(val_in) if (EDGE_SYM and dea_src and not need[1]) else None)
                    if gin is not None and dea_src:
                        gin = csr.from_source_order(gin)
                    g[1] = gin
            else:
                g[1] = dea if need[1] else None
        return tuple(g)
