"""Oracle (CPU, numpy, integer-exact) for the COO -> CSR index build.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  The reference never builds a CSR:
it hands ``edge_index`` [2,E] (row 0 = source, row 1 = target) to PyG
``propagate`` (/root/reference/libs/spect_conv.py:77), whose CPU scatter-add sums
the messages of one target in ascending EDGE order.  The HIP path instead walks
a CSR keyed by target; to keep the reference's summation order the edges of one
target must stay in input order -- i.e. a STABLE sort by target.  That is what
this restates, together with the source-keyed (transposed) view used by the
backward kernels.
"""
import numpy as np


def csr_from_coo(src, dst, num_nodes):
    """Stable counting sort of the edge list by ``dst``.

    Returns rowptr [N+1] int32, col [E] int32 (= src of each sorted edge),
    perm [E] int32 with sorted edge k == input edge perm[k].
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    perm = np.argsort(dst, kind='stable').astype(np.int32)
    counts = np.bincount(dst, minlength=num_nodes)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr.astype(np.int32), src[perm].astype(np.int32), perm


def transpose_view(src, dst, num_nodes, perm):
    """Source-keyed CSR over the same edges.

    Returns rowptr_t [N+1] int32, col_t [E] int32 (= dst), pos_t [E] int32 where
    pos_t[j] is the position, in the target-sorted order of ``csr_from_coo``, of
    the j-th source-sorted edge (values live once, in target-sorted order).
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    perm_t = np.argsort(src, kind='stable')
    counts = np.bincount(src, minlength=num_nodes)
    rowptr_t = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr_t[1:])
    inv = np.empty(len(src), dtype=np.int64)
    inv[np.asarray(perm, dtype=np.int64)] = np.arange(len(src))
    return rowptr_t.astype(np.int32), dst[perm_t].astype(np.int32), inv[perm_t].astype(np.int32)
