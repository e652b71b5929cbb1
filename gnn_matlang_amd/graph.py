"""Batch plumbing around the spectral layer: block-diagonal collate (the role PyG's DataLoader/Batch
plays in the reference scripts, e.g. /root/reference/Zinc12k.py:20-22) and the per-batch index
structure the HIP kernels walk.

``GraphCSR`` holds, on the device, for one ``edge_index`` [2,E] (row 0 = source, row 1 = target):
  rowptr/col/perm        edges stably sorted by TARGET  (forward: out[t] = sum over in-edges of t)
  rowptr_t/col_t/pos_t   edges stably sorted by SOURCE  (backward d/dX), pos_t = where that edge's
                         values live in target-sorted order (values are stored once)
It is built once per batch by the integer kernels of csrc/gml_csr.hip and cached on the
``edge_index`` tensor's identity, so the 4-5 layers of a model and every epoch re-use it.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor, got %s' % (name, type(t).__name__))
    if not t.is_cuda:
        raise RuntimeError('%s is on %s: the spectral layer runs only on an MI355X device through '
                           'libgml_hip.so; there is no CPU fallback' % (name, t.device))


class GraphCSR(object):
    __slots__ = ('N', 'E', 'device', 'rowptr', 'col', 'perm', 'rowptr_t', 'col_t', 'pos_t', 'perm_t', 'src_sorted',
                 '_ginfo', '_ginfo_t', '_gmax', '_gmax_t', 'ginfo_t128', 'gmax_t128', 'ginfo128', 'gmax128', 'tpos', '_val_cache', '_keep', '_r64',
                 '_r64t', '_bad', 'static_shape')

    def __init__(self):
        self._val_cache = OrderedDict()
        self._keep = None
        self._r64t = None
        self._r64 = None
        self._ginfo = self._ginfo_t = self._gmax = self._gmax_t = None
        self._bad = None
        self.static_shape = False                              # True: a static-shape batch whose tensors are refilled in place (dataset.py)

    @staticmethod
    def from_edge_index(edge_index, num_nodes, assume_source_sorted=True, static_caps=None):
        """assume_source_sorted: try the no-sort construction of the source view first (edge_index2 as SpectralDesign
        emits it is sorted by source, libs/utils.py:608-609); the kernel checks, and an unsorted input is rebuilt the
        general way after the one host read this function does anyway.
        static_caps = (max edges, max column window) per 128 source rows, known to the caller (dataset.DeviceDataset
        bounds them for a whole data set): NO host read happens -- the build is then a fixed sequence of launches that a HIP
        graph can capture and replay on new data of the same (padded) shape.  The caller vouches for sorted sources and
        ids in range; the kernels still clamp, so a violation cannot write out of bounds."""
        _require_cuda(edge_index, 'edge_index')
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError('edge_index must be int64 [2, E], got %s %s' % (edge_index.dtype, tuple(edge_index.shape)))
        ei = edge_index.contiguous()
        dev = ei.device
        E, N = int(ei.size(1)), int(num_nodes)
        g = GraphCSR()
        g.N, g.E, g.device = N, E, dev
        i32 = dict(dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            st = _stream(dev)
            L = _lib.lib()
            nws = max(int(L.gml_csr_workspace_bytes(N, E)), 4)
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)
            bad = ws[nws - 4:].view(torch.int32)               # bit 0: a node id out of range; bit 1: source keys not sorted
            bad.zero_()
            src, dst = ei[0], ei[1]
            g.rowptr, g.col, g.perm = torch.empty(N + 1, **i32), torch.empty(E, **i32), torch.empty(E, **i32)
            _lib.call('gml_csr_from_coo', _ptr(dst), _ptr(src), N, E, _ptr(g.rowptr), _ptr(g.col), _ptr(g.perm),
                      _ptr(ws), ws.numel(), st)
            g.rowptr_t, g.col_t, g.perm_t = torch.empty(N + 1, **i32), torch.empty(E, **i32), torch.empty(E, **i32)
            _lib.call('gml_csr_from_sorted_coo' if assume_source_sorted else 'gml_csr_from_coo', _ptr(src), _ptr(dst), N, E,
                      _ptr(g.rowptr_t), _ptr(g.col_t), _ptr(g.perm_t), _ptr(ws), ws.numel(), st)
            inv = torch.empty(E, **i32)
            g.pos_t = torch.empty(E, **i32)
            _lib.call('gml_csr_link_transpose', _ptr(g.perm), _ptr(g.perm_t), E, _ptr(inv), _ptr(g.pos_t), st)
            # tpos = inverse of pos_t: source-sorted position of every target-sorted edge
            g.tpos = torch.empty(E, **i32)
            _lib.call('gml_csr_link_transpose', _ptr(g.pos_t), _ptr(g.pos_t), E, _ptr(g.tpos), _ptr(inv), st)
            # 128-row group records of both views (the 8-wave kernels); the 64-row ones are built on first use
            ng2 = max((N + 127) // 128, 1)
            rec128 = int(_lib.lib().gml_csr_group_record_ints(128))
            g.ginfo_t128 = torch.zeros(ng2, rec128, **i32)
            g.ginfo128 = torch.zeros(ng2, rec128, **i32)
            _lib.call('gml_csr_group_info', _ptr(g.rowptr_t), _ptr(g.col_t), N, 128, _ptr(g.ginfo_t128), st)
            _lib.call('gml_csr_group_info', _ptr(g.rowptr), _ptr(g.col), N, 128, _ptr(g.ginfo128), st)
            # ONE device->host read per batch, at index-build time (not in the step): the per-batch maxima that size the LDS
            # staging of the fused backward, and the flag word of the index kernels (ids outside [0, num_nodes) were
            # clamped there, so nothing was written out of bounds: raise like the reference's scatter does)
            if static_caps is not None:
                g.gmax_t128 = (int(static_caps[0]), int(static_caps[1]))
                g.static_shape = True                          # contents change under the same tensors (HIP-graph replays): no data-dependent caches
                g.gmax128 = g.gmax_t128                        # (the caller bounds both views: symmetric masks)
                g.src_sorted = bool(assume_source_sorted)
                g._bad = bad.clone()                           # not read here (no host read): see check()
                return g
            mx = torch.stack([g.ginfo_t128[:, 1].max(), g.ginfo_t128[:, 3].max(), bad[0], g.ginfo128[:, 1].max(), g.ginfo128[:, 3].max()]).tolist()
            g.gmax_t128 = (int(mx[0]), int(mx[1]))
            g.gmax128 = (int(mx[3]), int(mx[4]))               # target view: largest group (edges, column window) of the forward kernels
            if mx[2] & 1:
                raise IndexError('edge_index holds node ids outside [0, %d)' % N)
            if mx[2] & 2:                                      # source keys were not sorted: general construction
                return GraphCSR.from_edge_index(edge_index, num_nodes, assume_source_sorted=False)
            g.src_sorted = bool(assume_source_sorted)
        return g

    def check(self):
        """Deferred validation of a CSR built with ``static_caps`` (the build itself reads nothing back, so that a HIP graph can
        capture it): raises if the index kernels flagged node ids out of range (they were clamped) or unsorted source keys
        (the source view then is NOT the one the kernels assume: gradients would be wrong).  One host read; call it once per
        epoch, or after replaying a captured step on new data."""
        bad = getattr(self, '_bad', None)
        if bad is None:
            return self
        v = int(bad.item())
        if v & 1:
            raise IndexError('edge_index holds node ids outside [0, %d)' % self.N)
        if v & 2:
            raise ValueError('edge_index is not sorted by source: a static-shape batch must be (SpectralDesign emits it so)')
        return self

    @staticmethod
    def _no_capture(what):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('%s needs a device->host read (per-batch group maxima): not possible while a HIP graph is being '
                               'captured -- this shape is outside the kernels served by static-shape batches' % what)

    def _need64(self):
        """64-row group records of both views (the 4-wave kernel families: shapes off the 8-wave kernels' set) + maxima"""
        if self._ginfo is None:
            self._no_capture('GraphCSR: 64-row group records')
            with torch.cuda.device(self.device):
                st = _stream(self.device)
                i32 = dict(dtype=torch.int32, device=self.device)
                ng = max((self.N + 63) // 64, 1)
                rec64 = int(_lib.lib().gml_csr_group_record_ints(64))
                gi, git = torch.zeros(ng, rec64, **i32), torch.zeros(ng, rec64, **i32)
                _lib.call('gml_csr_group_info', _ptr(self.rowptr), _ptr(self.col), self.N, 64, _ptr(gi), st)
                _lib.call('gml_csr_group_info', _ptr(self.rowptr_t), _ptr(self.col_t), self.N, 64, _ptr(git), st)
                mx = torch.stack([gi[:, 1].max(), gi[:, 3].max(), git[:, 1].max(), git[:, 3].max()]).tolist()
                self._gmax, self._gmax_t = (int(mx[0]), int(mx[1])), (int(mx[2]), int(mx[3]))
                self._ginfo, self._ginfo_t = gi, git

    ginfo = property(lambda self: (self._need64(), self._ginfo)[1])
    ginfo_t = property(lambda self: (self._need64(), self._ginfo_t)[1])
    gmax = property(lambda self: (self._need64(), self._gmax)[1])
    gmax_t = property(lambda self: (self._need64(), self._gmax_t)[1])

    def ranked64(self):
        """(records, (max edges, max window)) of the TARGET view in ranked 64-row groups: the staging schedule of the forward
        kernel's 4-wave geometry (GML_FWD_NW=4); built on first use."""
        if self._r64 is None:
            self._no_capture('GraphCSR: ranked 64-row group records')
            with torch.cuda.device(self.device):
                rec = int(_lib.lib().gml_csr_group_record_ints(_lib.GML_GROUPS64_RANKED))
                gi = torch.zeros(max((self.N + 63) // 64, 1), rec, dtype=torch.int32, device=self.device)
                _lib.call('gml_csr_group_info', _ptr(self.rowptr), _ptr(self.col), self.N, _lib.GML_GROUPS64_RANKED,
                          _ptr(gi), _stream(self.device))
                self._r64 = (gi, None)
        return self._r64

    def ranked64_t(self):
        """(records, (max edges, max window)) of the source view in ranked 64-row groups: the staging schedule of the
        4-wave backward kernel (two workgroups per CU); built on first use."""
        if self._r64t is None:
            self._no_capture('GraphCSR: ranked 64-row group records')
            with torch.cuda.device(self.device):
                rec = int(_lib.lib().gml_csr_group_record_ints(_lib.GML_GROUPS64_RANKED))
                gi = torch.zeros(max((self.N + 63) // 64, 1), rec, dtype=torch.int32, device=self.device)
                _lib.call('gml_csr_group_info', _ptr(self.rowptr_t), _ptr(self.col_t), self.N, _lib.GML_GROUPS64_RANKED,
                          _ptr(gi), _stream(self.device))
                mx = torch.stack([gi[:, 1].max(), gi[:, 3].max()]).tolist()
                self._r64t = (gi, (int(mx[0]), int(mx[1])))
        return self._r64t

    # values [E, S] in input-edge order -> target-sorted order (cached: raw supports are per-batch data)
    def sort_values(self, edge_attr, cache=True):
        key = (edge_attr.data_ptr(), edge_attr._version, tuple(edge_attr.shape))
        if cache and key in self._val_cache:
            return self._val_cache[key][1]
        ea = edge_attr.contiguous()
        out = torch.empty_like(ea)
        S = int(ea.size(1))
        # (the bf16 pre-split the edge kernels want belongs to the SOURCE-order copy -- the order the edge branch runs in
        #  when the layer trains, functional.ML3LayerFunction -- and is made by to_source_order / presplit)
        _lib.call('gml_gather_rows', _ptr(ea), _ptr(self.perm), _ptr(out), self.E, S, _stream(ea.device))
        if cache:
            self._val_cache[key] = (edge_attr, out)        # keep the source alive: its address is the key
            if self.src_sorted and ea is edge_attr:
                # input order IS source order: the source-order copy of these values is the input itself
                self._val_cache[('t', out.data_ptr(), out._version, tuple(out.shape))] = (out, ea)
            while len(self._val_cache) > 12:
                self._val_cache.popitem(last=False)
        return out

    def to_source_order(self, val_sorted, cache=False):
        """target-sorted [E,S] -> source-sorted (the order the backward kernel walks).  cache=True for
        per-batch data (the raw supports), keyed on the tensor's identity."""
        key = ('t', val_sorted.data_ptr(), val_sorted._version, tuple(val_sorted.shape))
        if cache and key in self._val_cache:
            return self._val_cache[key][1]
        out = torch.empty_like(val_sorted)
        S = int(val_sorted.size(1))
        if cache and S <= 8 and self.E > 0 and (S != 8 or (val_sorted.data_ptr() | out.data_ptr()) % 16 == 0):
            # per-batch data: the source-order rows and their bf16 pre-split (the matrix-core edge kernels' operand) in one pass
            es = torch.empty(self.E, 8, dtype=torch.int32, device=out.device)
            _lib.call('gml_gather_rows_presplit', _ptr(val_sorted), _ptr(self.pos_t), _ptr(out), _ptr(es), self.E, S,
                      _stream(out.device))
            self._val_cache[('p', out.data_ptr(), out._version, tuple(out.shape))] = (out, es)
        else:
            _lib.call('gml_gather_rows', _ptr(val_sorted), _ptr(self.pos_t), _ptr(out), self.E, S, _stream(out.device))
        if cache:
            self._val_cache[key] = (val_sorted, out)
            while len(self._val_cache) > 12:
                self._val_cache.popitem(last=False)
        return out

    def presplit(self, val):
        """bf16 hi | lo image of a per-batch value array (cached on the tensor's identity like the other derived
        arrays); None when the edge kernels do not use one (S > 16)."""
        from .functional import edge_presplit
        if val.size(1) > 16:
            return None
        if val.requires_grad:                                  # trained supports change every step: split, do not cache
            return edge_presplit(val.detach())
        key = ('p', val.data_ptr(), val._version, tuple(val.shape))
        hit = self._val_cache.get(key)
        if hit is not None:
            return hit[1]
        out = edge_presplit(val)
        if out is not None:
            self._val_cache[key] = (val, out)
            while len(self._val_cache) > 12:
                self._val_cache.popitem(last=False)
        return out

    def sym_index(self, val, view='source'):
        """(uid, mir) int32 [U] for per-batch supports in SOURCE order (view='source': the order the training edge branch runs in) or in
        target-sorted order (view='target': inference): the edges the edge branch has to evaluate and, per entry, the mirror edge (j, i)
        that carries bitwise the same row (-1: none) -- include/gml.h gml_edge_sym_flags.  Cached on the tensor's identity like the
        other derived arrays; None when nothing can be shared (S outside 2 .. 16, supports that carry a gradient, fewer than 10 % of the
        evaluations saved), for static-shape batches (their tensors are refilled in place by every replay of a captured step) or
        while a HIP graph is being captured (the list's length is data)."""
        S = int(val.size(1))
        if not (2 <= S <= 16) or val.requires_grad or self.E == 0 or self.E * S * 4 >= 0x7fffff00 or getattr(self, 'static_shape', False):
            return None
        key = ('y' + view[0], val.data_ptr(), val._version, tuple(val.shape))
        hit = self._val_cache.get(key)
        if hit is not None:
            return hit[1]
        if torch.cuda.is_current_stream_capturing():
            return None
        rp, cl = (self.rowptr_t, self.col_t) if view == 'source' else (self.rowptr, self.col)
        flag = torch.empty(self.E, dtype=torch.int32, device=val.device)
        mirror = torch.empty(self.E, dtype=torch.int32, device=val.device)
        _lib.call('gml_edge_sym_flags', _ptr(rp), _ptr(cl), _ptr(val), self.N, self.E, S, _ptr(flag), _ptr(mirror), _stream(val.device))
        idx = torch.nonzero(flag, as_tuple=False).view(-1)
        out = (idx.to(torch.int32), mirror[idx]) if idx.numel() <= 0.9 * self.E else None
        self._val_cache[key] = (val, out)
        while len(self._val_cache) > 12:
            self._val_cache.popitem(last=False)
        return out

    def from_source_order(self, val_t):
        """source-sorted [E,S] -> target-sorted."""
        out = torch.empty_like(val_t)
        _lib.call('gml_scatter_rows', _ptr(val_t), _ptr(self.pos_t), _ptr(out), self.E, int(val_t.size(1)),
                  _stream(val_t.device))
        return out

    def unsort_values(self, val_sorted):
        """target-sorted [E,S] -> input-edge order (gradient of sort_values)."""
        out = torch.empty_like(val_sorted)
        _lib.call('gml_scatter_rows', _ptr(val_sorted), _ptr(self.perm), _ptr(out), self.E, int(val_sorted.size(1)),
                  _stream(val_sorted.device))
        return out


_CSR_CACHE = OrderedDict()
_CSR_CACHE_MAX = 8


def csr_for(edge_index, num_nodes):
    """GraphCSR for this edge_index (or pass-through if one is given)."""
    if isinstance(edge_index, GraphCSR):
        if edge_index.N != num_nodes:
            raise ValueError('GraphCSR was built for %d nodes, x has %d rows' % (edge_index.N, num_nodes))
        return edge_index
    _require_cuda(edge_index, 'edge_index')
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes), str(edge_index.device))
    hit = _CSR_CACHE.get(key)
    if hit is not None:
        _CSR_CACHE.move_to_end(key)
        return hit
    g = GraphCSR.from_edge_index(edge_index, num_nodes)
    g._keep = edge_index
    _CSR_CACHE[key] = g
    while len(_CSR_CACHE) > _CSR_CACHE_MAX:
        _CSR_CACHE.popitem(last=False)
    return g


def clear_caches():
    _CSR_CACHE.clear()


# --------------------------------------------------------------------------------------------
# collate: list of graphs -> one block-diagonal batch (PyG-compatible field names)
# --------------------------------------------------------------------------------------------
class Batch(object):
    """x [N,F], edge_index [2,e], edge_index2 [2,E], edge_attr2 [E,S], batch [N], ptr [B+1], y [B]."""

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self._csr = {}

    @property
    def num_graphs(self):
        return int(self.ptr.numel() - 1)

    def to(self, device):
        out = Batch(**{k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v)
                       for k, v in self.__dict__.items() if not k.startswith('_')})
        return out

    def csr(self, which='edge_index2'):
        """GraphCSR of ``edge_index2`` (spectral supports) or ``edge_index`` (raw adjacency), built once."""
        if which not in self._csr:
            caps = getattr(self, 'static_caps', None) if which == 'edge_index2' else None
            self._csr[which] = GraphCSR.from_edge_index(getattr(self, which), int(self.x.size(0)), static_caps=caps)
        return self._csr[which]


def collate(graphs):
    """graphs: iterable of dicts with x [n,f], edge_index [2,e] and optionally edge_index2/edge_attr2/y
    (numpy or torch).  Node ids are offset per graph; everything stays on the host."""
    graphs = list(graphs)
    xs, e1, e2, ea, bt, ys, ptr, off = [], [], [], [], [], [], [0], 0
    for g, d in enumerate(graphs):
        x = np.asarray(d['x'], dtype=np.float32)
        n = x.shape[0]
        xs.append(x)
        e1.append(np.asarray(d['edge_index'], dtype=np.int64) + off)
        if 'edge_index2' in d:
            e2.append(np.asarray(d['edge_index2'], dtype=np.int64) + off)
            ea.append(np.asarray(d['edge_attr2'], dtype=np.float32))
        bt.append(np.full(n, g, dtype=np.int64))
        ys.append(d.get('y', 0))
        off += n
        ptr.append(off)
    out = dict(x=torch.from_numpy(np.concatenate(xs)), edge_index=torch.from_numpy(np.concatenate(e1, 1)),
               batch=torch.from_numpy(np.concatenate(bt)), ptr=torch.tensor(ptr, dtype=torch.int32),
               y=torch.tensor(np.asarray(ys)))
    if e2:
        out['edge_index2'] = torch.from_numpy(np.concatenate(e2, 1))
        out['edge_attr2'] = torch.from_numpy(np.concatenate(ea))
    return Batch(**out)


def shard_graphs(num_graphs, rank, world_size):
    """Contiguous, balanced split of graph ids over ranks (data parallel over graphs)."""
    base, rem = divmod(num_graphs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_graphs_balanced(work, rank, world_size):
    """Contiguous split of the graphs [0, len(work)) over ranks with near-equal TOTAL work instead of equal counts
    (balance by the sum of nnz when graph sizes vary, e.g. proteins' 4 .. 620 nodes): rank r takes the graphs
    whose cumulative work lies in (r, r + 1] x total / world_size.  ``work``: per-graph cost (support edges).
    Every rank computes the same cut points, and their ranges tile [0, G) without gaps or overlap."""
    w = np.asarray(work, dtype=np.float64)
    G = int(w.size)
    if G == 0:
        return 0, 0
    cum = np.cumsum(w)
    total = float(cum[-1])
    if total <= 0:
        return shard_graphs(G, rank, world_size)
    mid = cum - 0.5 * w                                       # a graph belongs to the rank its mid-point falls into
    cuts = np.searchsorted(mid, total * np.arange(1, world_size) / world_size, side='left')
    cuts = np.concatenate([[0], cuts, [G]])
    return int(cuts[rank]), int(cuts[rank + 1])
