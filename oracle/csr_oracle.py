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


def group_records(rowptr, col, num_rows, group_rows):
    """Per-group staging records of gml_csr_group_info (include/gml.h): {first edge, #edges, smallest column id,
    column-window width}; for 128-row groups also the degree-ranked row order the backward kernel's lane positions use:
    rows ranked by (degree descending, index ascending), rows past the end last; rank block `a` goes to wave a's
    position block and block 7-a to wave a+4's, with a = ((wave & 3) + group) & 3."""
    rowptr, col = np.asarray(rowptr), np.asarray(col)
    ng = max((num_rows + group_rows - 1) // group_rows, 1) if num_rows > 0 else 0
    info = np.zeros((ng, 4), np.int64)
    order = np.zeros((ng, group_rows), np.int64)
    for g in range(ng):
        r0, r1 = g * group_rows, min((g + 1) * group_rows, num_rows)
        kb, ke = int(rowptr[r0]), int(rowptr[r1])
        c = col[kb:ke]
        info[g] = (kb, ke - kb, c.min() if ke > kb else 0, c.max() - c.min() + 1 if ke > kb else 0)
        deg = np.full(group_rows, -1, np.int64)
        deg[:r1 - r0] = rowptr[r0 + 1:r1 + 1] - rowptr[r0:r1]
        row_of_rank = np.lexsort((np.arange(group_rows), -deg))
        for pos in range(group_rows):
            wave, i = pos >> 4, pos & 15
            if group_rows == 128:
                a = ((wave & 3) + g) & 3
                blk = a if wave < 4 else 7 - a
                order[g, pos] = row_of_rank[blk * 16 + i]
            else:
                order[g, pos] = pos
    return info, order
