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

    def epoch(self, batch_size, generator=None, shuffle=True):
        """yields one shuffled epoch of batches (the DataLoader(shuffle=True) loop of Zinc12k.py:20,359)."""
        G = len(self)
        dev = self.node_ptr.device
        perm = torch.randperm(G, generator=generator).to(dev) if shuffle else torch.arange(G, device=dev)
        for i in range(0, G, batch_size):
            yield self.batch(perm[i:i + batch_size])
