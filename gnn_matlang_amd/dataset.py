"""Device-resident data set and batch assembly (the role ``InMemoryDataset`` + ``DataLoader`` play in the reference
scripts, e.g. /root/reference/Zinc12k.py:13-22, where every batch is collated on the host and copied with
``data.to(device)``, Zinc12k.py:360).

Here the whole data set -- node features, raw edges, the structural edges of the mask and their ``m x S`` supports
(``SpectralDesign``'s output) -- is concatenated once and kept in HBM (ZINC-12k with its supports is ~110 MB); a batch is
assembled ON the device from a tensor of graph ids with a handful of gathers, so an epoch of shuffled mini-batches never
touches the host arrays again.  The result is a ``Batch`` with PyG-compatible field names.
"""
import numpy as np
import torch

from .graph import Batch


class DeviceDataset(object):
    """x [N,F], node_ptr [G+1], edge_index / edge_index2 (graph-LOCAL node ids) with edge_ptr / edge_ptr2 [G+1],
    edge_attr2 [E2,S], y [G]; all on one device."""

    def __init__(self, x, node_ptr, edge_index, edge_ptr, edge_index2, edge_ptr2, edge_attr2, y):
        self.x, self.node_ptr, self.edge_index, self.edge_ptr = x, node_ptr, edge_index, edge_ptr
        self.edge_index2, self.edge_ptr2, self.edge_attr2, self.y = edge_index2, edge_ptr2, edge_attr2, y

    def __len__(self):
        return int(self.node_ptr.numel() - 1)

    @staticmethod
    def from_graphs(graphs, device):
        """graphs: dicts with x, edge_index, y and (optionally) edge_index2 / edge_attr2, as ``SpectralDesign`` returns."""
        graphs = list(graphs)
        xs = [np.asarray(g['x'], dtype=np.float32) for g in graphs]
        nptr = np.zeros(len(graphs) + 1, dtype=np.int64)
        nptr[1:] = np.cumsum([x.shape[0] for x in xs])

        def edges(key):
            es = [np.asarray(g[key], dtype=np.int64) for g in graphs]
            ptr = np.zeros(len(graphs) + 1, dtype=np.int64)
            ptr[1:] = np.cumsum([e.shape[1] for e in es])
            return torch.from_numpy(np.concatenate(es, 1)).to(device), torch.from_numpy(ptr).to(device)
        ei, ep = edges('edge_index')
        if 'edge_index2' in graphs[0]:
            ei2, ep2 = edges('edge_index2')
            ea2 = torch.from_numpy(np.concatenate([np.asarray(g['edge_attr2'], dtype=np.float32) for g in graphs])).to(device)
        else:
            ei2 = ep2 = ea2 = None
        y = torch.tensor(np.asarray([g.get('y', 0) for g in graphs])).to(device)
        return DeviceDataset(torch.from_numpy(np.concatenate(xs)).to(device), torch.from_numpy(nptr).to(device),
                             ei, ep, ei2, ep2, ea2, y)

    @staticmethod
    def _ranges(ptr, ids):
        """(flat source positions of the segments ids select, new segment pointer [B+1]); one host read (the total)."""
        lo, n = ptr[ids], ptr[ids + 1] - ptr[ids]
        newptr = torch.zeros(ids.numel() + 1, dtype=torch.int64, device=ids.device)
        newptr[1:] = torch.cumsum(n, 0)
        total = int(newptr[-1])
        seg = torch.repeat_interleave(torch.arange(ids.numel(), device=ids.device), n, output_size=total)
        pos = torch.arange(total, device=ids.device) - newptr[seg] + lo[seg]
        return pos, newptr, seg

    def batch(self, ids):
        """Block-diagonal batch of the graphs ``ids`` (int64 tensor on the data set's device), in that order."""
        npos, nptr, nseg = self._ranges(self.node_ptr, ids)
        out = dict(x=self.x[npos], batch=nseg, ptr=nptr.int(), y=self.y[ids])
        base = nptr[:-1]

        def take(ei, ep):
            epos, _, eseg = self._ranges(ep, ids)
            return ei[:, epos] + base[eseg].unsqueeze(0), epos
        out['edge_index'], _ = take(self.edge_index, self.edge_ptr)
        if self.edge_index2 is not None:
            out['edge_index2'], epos2 = take(self.edge_index2, self.edge_ptr2)
            out['edge_attr2'] = self.edge_attr2[epos2]
        return Batch(**out)

    # ------------------------------------------------------------------ static shapes: one captured step for every batch
    def bounds(self, batch_size):
        """dict(n_pad, e2_pad, dmax, deal, caps) that hold for EVERY batch of batch_size graphs of this data set -- what a
        HIP-graph-captured step is sized with (one host read per data set, not per batch).  caps = (edges, column window)
        per 128 source rows; n_pad leaves room for e2_pad / deal padding nodes, so that the padding edges can be dealt `deal`
        (<= dmax) per node and the caps also hold on the padding."""
        n = (self.node_ptr[1:] - self.node_ptr[:-1])
        e = (self.edge_ptr2[1:] - self.edge_ptr2[:-1])
        gid = torch.repeat_interleave(torch.arange(len(self), device=n.device), e)
        deg = torch.bincount(self.edge_index2[0] + self.node_ptr[gid], minlength=int(self.x.size(0)))
        nmax, n_top, e_top, dmax = [int(v) for v in torch.stack([
            n.max(), torch.topk(n, min(batch_size, n.numel()))[0].sum(), torch.topk(e, min(batch_size, e.numel()))[0].sum(),
            deg.max()]).tolist()]
        dmax = max(dmax, 1)
        e2_pad = (e_top + 63) // 64 * 64
        # padding edges are dealt `deal` per padding node = the data set's mean degree, rounded up: a 128-row group of padding then
        # carries the edge count of a group of real rows (dealt dmax per node -- rounds 2-3 -- the padding groups held 2.2 x the edges
        # of a real group and a one-group-per-workgroup launch waited for them: ZINC batch 64, conv forward 52 -> 23 us)
        deal = max(1, min(dmax, -(-int(self.edge_index2.size(1)) // max(int(self.x.size(0)), 1))))
        n_pad = (n_top + (e2_pad + deal - 1) // deal + 127) // 128 * 128
        return dict(n_pad=n_pad, e2_pad=e2_pad, dmax=dmax, deal=deal, caps=(128 * dmax, 128 + 2 * nmax))

    def batch_padded(self, ids, bounds):
        """The batch of graphs ``ids`` (entries equal to len(self) = no graph) padded to bounds['n_pad'] nodes and
        bounds['e2_pad'] support edges with torch ops of STATIC shapes only (no host read): padding nodes carry zero
        features and form one extra graph (index B) at the end; padding edges are zero-valued self loops dealt bounds['deal'] per
        padding node (sorted by source like the rest; a zero support stays zero through the bias-free edge MLP and moves
        no gradient).  Returns a Batch with ptr [B + 2], y [B + 1] and ``graph_valid`` [B] (0 for absent graphs)."""
        n_pad, e2_pad, dmax = bounds['n_pad'], bounds['e2_pad'], bounds.get('deal', bounds['dmax'])
        dev = ids.device
        B = int(ids.numel())
        G = len(self)
        has = ids < G
        idc = ids.clamp(max=G - 1)

        def layout(ptr, total_pad):
            lo = ptr[idc]
            cnt = torch.where(has, ptr[idc + 1] - lo, torch.zeros_like(lo))
            new = torch.zeros(B + 1, dtype=torch.int64, device=dev)
            new[1:] = torch.cumsum(cnt, 0)
            j = torch.arange(total_pad, device=dev)
            seg = torch.searchsorted(new[1:].contiguous(), j, right=True)      # = B at and beyond the real total
            ok = seg < B
            sc = seg.clamp(max=B - 1)
            return torch.where(ok, j - new[sc] + lo[sc], torch.zeros_like(j)), new, seg, ok, sc, j
        npos, nptr, nseg, nok, _, _ = layout(self.node_ptr, n_pad)
        epos, eptr, _, eok, esc, k = layout(self.edge_ptr2, e2_pad)
        x = self.x[npos] * nok.unsqueeze(1).to(self.x.dtype)
        pad_node = (nptr[-1] + (k - eptr[-1]).clamp(min=0) // dmax).clamp(max=n_pad - 1)
        ei2 = torch.where(eok.unsqueeze(0), self.edge_index2[:, epos] + nptr[esc].unsqueeze(0), pad_node.unsqueeze(0))
        ea2 = self.edge_attr2[epos] * eok.unsqueeze(1).to(self.edge_attr2.dtype)
        ptr = torch.cat([nptr, torch.full((1,), n_pad, dtype=torch.int64, device=dev)]).int()
        y = torch.cat([torch.where(has, self.y[idc], torch.zeros_like(self.y[idc])), torch.zeros(1, dtype=self.y.dtype, device=dev)])
        # (no `edge_index`: the raw adjacency is not padded here -- a consumer of data.csr('edge_index'), e.g. a GNNML1 model or a
        #  K = 1 raw-adjacency conv, must fail loudly on a padded batch instead of computing on the support edges)
        b = Batch(x=x, edge_index2=ei2, edge_attr2=ea2, batch=nseg, ptr=ptr, y=y, graph_valid=has.to(self.x.dtype))
        b.static_caps = bounds['caps']
        b.pad_graph = True                                 # the last graph is padding: pooling skips it (its pooled row is zero)
        return b

    # ------------------------------------------------------------------ precomputed per-graph structure: a batch in ONE launch
    def prepare(self):
        """Once per data set: every graph's own index structure -- the stable target sort of its (source-sorted) support edges,
        its inverse, both local row-pointer prefixes -- and the bf16 pre-split of all supports.  A batch is the block-diagonal
        union of graphs whose structure never changes, so ``batch_assembled`` only adds offsets (csrc/gml_csr.hip
        gml_batch_assemble).  Returns self."""
        if getattr(self, '_prep', None) is not None:
            return self
        from . import functional as Fn
        dev = self.x.device
        E2, Nall, G = int(self.edge_index2.size(1)), int(self.x.size(0)), len(self)
        if self.y.dtype != torch.float32 or self.x.dtype != torch.float32:
            raise TypeError('prepare(): float32 features and targets')
        e = self.edge_ptr2[1:] - self.edge_ptr2[:-1]
        gid = torch.repeat_interleave(torch.arange(G, device=dev), e, output_size=E2)
        nbase, ebase = self.node_ptr[gid], self.edge_ptr2[gid]
        src, dst = self.edge_index2[0] + nbase, self.edge_index2[1] + nbase
        if E2 > 1 and not bool((src[1:] >= src[:-1]).all()):
            raise ValueError('prepare(): the support edges of every graph must be sorted by source (SpectralDesign emits them so)')
        order = torch.sort(dst, stable=True)[1]                           # global stable target sort = per-graph stable target sort
        k = torch.arange(E2, device=dev)
        tperm = (order - ebase[order]).int()                              # [position in target order] -> local source-order position
        tinv = torch.empty(E2, dtype=torch.int32, device=dev)
        tinv[order] = (k - ebase[order]).int()                            # [source-order position] -> local position in target order
        first = self.edge_ptr2[torch.repeat_interleave(torch.arange(G, device=dev), self.node_ptr[1:] - self.node_ptr[:-1], output_size=Nall)]

        def local_rows(keys):
            cnt = torch.bincount(keys, minlength=Nall)
            return (torch.cumsum(cnt, 0) - cnt - first).int()
        S = int(self.edge_attr2.size(1))
        es = Fn.edge_presplit(self.edge_attr2.contiguous()) if S <= 8 else None
        self._prep = dict(tperm=tperm.contiguous(), tinv=tinv, rp_src=local_rows(src), rp_dst=local_rows(dst), es=es,
                          x=self.x.contiguous(), ea=self.edge_attr2.contiguous(), ei2=self.edge_index2.contiguous(), y=self.y.contiguous())
        return self

    def batch_assembled(self, ids, bounds):
        """``batch_padded(ids, bounds)`` AND its index structure (Batch.csr('edge_index2')) in one kernel launch plus the two
        group-record passes, from the per-graph structure ``prepare()`` computed once: bit-identical tensors and CSR arrays
        (tests/test_gpu_parity.py), no host read, capturable."""
        from . import _lib
        from .graph import GraphCSR, _ptr, _stream
        self.prepare()
        P = self._prep
        n_pad, e2_pad, dmax = bounds['n_pad'], bounds['e2_pad'], bounds.get('deal', bounds['dmax'])
        dev = ids.device
        B, F, S = int(ids.numel()), int(self.x.size(1)), int(self.edge_attr2.size(1))
        if ids.dtype != torch.int64 or not ids.is_contiguous():
            raise ValueError('ids: contiguous int64')
        f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
        ldx = (F + 3) // 4 * 4                                         # float4-addressable rows: the first layer reads them without a padding copy
        xbuf, ea = torch.empty(n_pad, ldx, **f32), torch.empty(e2_pad, S, **f32)
        x = xbuf[:, :F]
        es = torch.empty(e2_pad, 8, **i32) if P['es'] is not None else None
        y, valid = torch.empty(B + 1, **f32), torch.empty(B, **f32)
        ptr, batch = torch.empty(B + 2, **i32), torch.empty(n_pad, **i32)
        g = GraphCSR()
        g.N, g.E, g.device = n_pad, e2_pad, dev
        g.rowptr, g.col, g.perm = torch.empty(n_pad + 1, **i32), torch.empty(e2_pad, **i32), torch.empty(e2_pad, **i32)
        g.rowptr_t, g.col_t, g.pos_t = torch.empty(n_pad + 1, **i32), torch.empty(e2_pad, **i32), torch.empty(e2_pad, **i32)
        ident = P.get('ident')
        if ident is None or ident.numel() != e2_pad:
            ident = P['ident'] = torch.arange(e2_pad, **i32)
        g.perm_t, g.tpos = ident, g.perm
        d = _lib.BatchDesc()
        for name, t in (('node_ptr', self.node_ptr), ('edge_ptr2', self.edge_ptr2), ('x', P['x']), ('edge_index2', P['ei2']), ('edge_attr2', P['ea']),
                        ('es', P['es']), ('tperm', P['tperm']), ('tinv', P['tinv']), ('rp_src', P['rp_src']), ('rp_dst', P['rp_dst']), ('y', P['y']),
                        ('ids', ids), ('x_out', xbuf), ('ea_out', ea), ('es_out', es), ('y_out', y), ('valid_out', valid), ('ptr_out', ptr),
                        ('batch_out', batch), ('rowptr', g.rowptr), ('col', g.col), ('perm', g.perm), ('rowptr_t', g.rowptr_t),
                        ('col_t', g.col_t), ('pos_t', g.pos_t)):
            setattr(d, name, _ptr(t) if t is not None else None)
        d.G, d.E2all, d.F, d.S, d.B, d.n_pad, d.e2_pad, d.dmax = len(self), int(self.edge_index2.size(1)), F, S, B, n_pad, e2_pad, dmax
        d.ldx_out = ldx
        with torch.cuda.device(dev):
            st = _stream(dev)
            import ctypes
            ng2 = max((n_pad + 127) // 128, 1)
            rec128 = int(_lib.lib().gml_csr_group_record_ints(128))
            both = torch.empty(2, ng2, rec128, **i32)                      # (every int of a 128-row record is written: no fill)
            g.ginfo_t128, g.ginfo128 = both[0], both[1]
            d.ginfo128, d.ginfo_t128 = _ptr(g.ginfo128), _ptr(g.ginfo_t128)  # round 5: the group records of both views come out of the same launch
            _lib.call('gml_batch_assemble', ctypes.addressof(d), st)
        g.gmax_t128 = g.gmax128 = (int(bounds['caps'][0]), int(bounds['caps'][1]))
        g.src_sorted = True
        g.static_shape = True
        if es is not None:                                                # the pre-split supports travel with the batch (functional.presplit_of)
            g._val_cache[('p', ea.data_ptr(), ea._version, tuple(ea.shape))] = (ea, es)
        b = Batch(x=x, edge_attr2=ea, batch=batch, ptr=ptr, y=y, graph_valid=valid)
        b.static_caps = bounds['caps']
        b.pad_graph = True
        b._csr['edge_index2'] = g
        b._batch_i32 = batch
        return b

    def epoch_static(self, batch_size, generator=None, shuffle=True, bounds=None):
        """One shuffled epoch as STATIC-shape batches (``batch_assembled``): no host read per batch, every batch of the same padded
        shape -- absent slots in the last one.  Use the loss form of a padded batch: ``((pre[:B, 0] - b.y[:B]).abs() * b.graph_valid).sum()``
        (Zinc12k.py:365's L1 sum over the real graphs)."""
        G = len(self)
        dev = self.node_ptr.device
        bd = bounds if bounds is not None else self.bounds(batch_size)
        perm = torch.randperm(G, generator=generator).to(dev) if shuffle else torch.arange(G, device=dev)
        perm = torch.cat([perm, torch.full(((-G) % batch_size,), G, dtype=torch.int64, device=dev)])
        for i in range(0, perm.numel(), batch_size):
            yield self.batch_assembled(perm[i:i + batch_size].contiguous(), bd)

    def epoch(self, batch_size, generator=None, shuffle=True):
        """yields one shuffled epoch of batches (the DataLoader(shuffle=True) loop of Zinc12k.py:20,359)."""
        G = len(self)
        dev = self.node_ptr.device
        perm = torch.randperm(G, generator=generator).to(dev) if shuffle else torch.arange(G, device=dev)
        for i in range(0, G, batch_size):
            yield self.batch(perm[i:i + batch_size])
