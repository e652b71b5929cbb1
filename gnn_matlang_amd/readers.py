"""Raw data readers of the two file formats the reference's GNNML data sets come in, numpy only.

The reference loads them through scipy (``sio.loadmat``, libs/utils.py:195) and networkx (``nx.read_graph6``,
libs/utils.py:509) inside PyG ``InMemoryDataset.process`` bodies; neither PyG nor the dataset classes exist here
(SURVEY D7), so the product carries its own readers and the two ``process`` bodies the BASELINE configs use:

  read_mat      MATLAB Level-5 .mat: numeric / char / cell / struct / sparse arrays, compressed elements (zlib)
  read_graph6   graph6 text (one graph per line, optional >>graph6<< header)
  load_mutag    libs/utils.py:192-209  (A, F, y of dataset/mutag/raw/mutag.mat -> x, edge_index, y per graph)
  load_sr       libs/utils.py:506-513  (sr251256.g6 -> x = ones, undirected edge_index, y = 0)

Each returns graphs as (x [n, f] float32, edge_index [2, e] int64 in row-major ``np.where`` order, y), the input format
of ``SpectralDesign.design_many`` / ``graph.collate``.
"""
import struct
import zlib

import numpy as np

_MI = {1: ('b', 1), 2: ('B', 1), 3: ('h', 2), 4: ('H', 2), 5: ('i', 4), 6: ('I', 4), 7: ('f', 4), 9: ('d', 8),
       12: ('q', 8), 13: ('Q', 8), 16: ('B', 1), 17: ('H', 2), 18: ('I', 4)}
_MI_MATRIX, _MI_COMPRESSED = 14, 15
_MX_CELL, _MX_STRUCT, _MX_OBJECT, _MX_CHAR, _MX_SPARSE = 1, 2, 3, 4, 5


class _Reader(object):
    def __init__(self, buf, order):
        self.b, self.o, self.p = buf, order, 0

    def tag(self):
        """(type, nbytes, data offset, offset of the next element); handles the small-element form"""
        t, = struct.unpack_from(self.o + 'I', self.b, self.p)
        if t >> 16:                                           # small data element: 2-byte size, 2-byte type, 4 data bytes
            return t & 0xffff, t >> 16, self.p + 4, self.p + 8
        n, = struct.unpack_from(self.o + 'I', self.b, self.p + 4)
        return t, n, self.p + 8, self.p + 8 + ((n + 7) // 8) * 8 if t != _MI_COMPRESSED else self.p + 8 + n

    def numeric(self):
        t, n, d, nxt = self.tag()
        self.p = nxt
        code, size = _MI[t]
        return np.frombuffer(self.b, dtype=np.dtype(code).newbyteorder(self.o), count=n // size, offset=d)

    def element(self):
        """next top-level or nested element -> (name, value)"""
        t, n, d, nxt = self.tag()
        if t == _MI_COMPRESSED:
            sub = _Reader(zlib.decompress(self.b[d:d + n]), self.o)
            self.p = nxt
            return sub.element()
        if t != _MI_MATRIX:
            self.p = nxt
            return None, None
        end = nxt
        if n == 0:                                            # empty matrix placeholder
            self.p = end
            return '', np.zeros((0, 0))
        self.p = d
        flags = self.numeric()
        cls, is_complex = int(flags[0]) & 0xff, bool(int(flags[0]) & 0x800)
        dims = [int(v) for v in self.numeric()]
        name = self.numeric().tobytes().decode('latin1')
        if cls == _MX_CELL:
            out = np.empty(int(np.prod(dims)), dtype=object)
            for i in range(out.size):
                out[i] = self.element()[1]
            val = out.reshape(dims, order='F')
        elif cls in (_MX_STRUCT, _MX_OBJECT):
            if cls == _MX_OBJECT:
                self.numeric()
            flen = int(self.numeric()[0])
            raw = self.numeric().tobytes()
            fields = [raw[i:i + flen].split(b'\0', 1)[0].decode('latin1') for i in range(0, len(raw), flen)]
            cnt = int(np.prod(dims))
            recs = [{f: self.element()[1] for f in fields} for _ in range(cnt)]
            val = recs[0] if cnt == 1 else np.array(recs, dtype=object).reshape(dims, order='F')
        elif cls == _MX_SPARSE:
            ir, jc = self.numeric().astype(np.int64), self.numeric().astype(np.int64)
            re = self.numeric()
            val = np.zeros(dims, dtype=np.float64 if re.dtype.kind == 'f' else re.dtype)
            for c in range(dims[1]):
                val[ir[jc[c]:jc[c + 1]], c] = re[jc[c]:jc[c + 1]]
        else:
            re = self.numeric()
            if is_complex:
                re = re + 1j * self.numeric()
            if cls == _MX_CHAR:
                chars = ''.join(chr(int(c)) for c in re)
                val = chars if len(dims) == 2 and dims[0] <= 1 else np.array(list(chars)).reshape(dims, order='F')
            else:
                val = np.array(re).reshape(dims, order='F')   # MATLAB stores column-major
        self.p = end
        return name, val


def read_mat(path):
    """dict name -> array of a MATLAB 5.0 MAT-file (the subset above; what ``scipy.io.loadmat`` returns for it, with cell
    arrays as object arrays)."""
    buf = open(path, 'rb').read()
    if len(buf) < 128 or not buf.startswith(b'MATLAB 5.0 MAT-file'):
        raise ValueError('%s is not a Level-5 MAT-file' % path)
    order = '<' if buf[126:128] == b'IM' else '>'
    r = _Reader(buf, order)
    r.p = 128
    out = {}
    while r.p + 8 <= len(buf):
        name, val = r.element()
        if name is not None:
            out[name] = val
    return out


def read_graph6(path):
    """list of symmetric 0/1 adjacency matrices (uint8) of a graph6 file (n < 258048)"""
    graphs = []
    for line in open(path, 'rb').read().split(b'\n'):
        line = line.strip()
        if line.startswith(b'>>graph6<<'):
            line = line[10:]
        if not line:
            continue
        d = np.frombuffer(line, dtype=np.uint8).astype(np.int64) - 63
        if d.min() < 0 or d.max() > 63:
            raise ValueError('%s: not graph6 text' % path)
        if d[0] <= 62:
            n, d = int(d[0]), d[1:]
        elif d[1] <= 62:
            n, d = int((d[1] << 12) | (d[2] << 6) | d[3]), d[4:]
        else:
            raise ValueError('%s: graphs of this size are not supported' % path)
        bits = ((d[:, None] >> np.arange(5, -1, -1)) & 1).reshape(-1)          # 6 bits per character, high bit first
        iu = np.triu_indices(n, 1)
        order = np.lexsort((iu[0], iu[1]))                                     # bit k = x(i, j), columns first: (0,1),(0,2),(1,2),...
        A = np.zeros((n, n), dtype=np.uint8)
        A[iu[0][order], iu[1][order]] = bits[:order.size]
        graphs.append(A | A.T)
    return graphs


def _edges(A):
    r, c = np.where(A > 0)
    return np.vstack((r, c)).astype(np.int64)


def load_mutag(path):
    """libs/utils.py:192-209: graphs of mutag.mat as (x, edge_index, y), y = (label + 1) // 2"""
    a = read_mat(path)
    A, F = a['A'].reshape(-1), a['F'].reshape(-1)
    Y = ((a['y'].astype(np.int64) + 1) // 2).astype(np.float32).reshape(-1)
    return [(np.asarray(F[i], dtype=np.float32), _edges(A[i]), np.float32(Y[i])) for i in range(A.size)]


def load_sr(path):
    """libs/utils.py:506-513: x = ones [n, 1], symmetric edges, y = 0 (sr25.py, graph8c.py)"""
    return [(np.ones((A.shape[0], 1), dtype=np.float32), _edges(A), np.float32(0)) for A in read_graph6(path)]
