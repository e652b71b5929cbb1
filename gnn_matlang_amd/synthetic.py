"""Seeded synthetic graph workloads shaped like the BASELINE configs (host side, numpy).

The real ZINC / counting / MNIST blobs are absent from the reference tree
(/root/reference/.MISSING_LARGE_BLOBS) and there is no network, so benches and
tests use these generators:

  Z  ZINC-12k-like  : n ~ clip(round(N(23,4)), 9, 37), random tree + ~0.08 n ring
                      closures, max degree 4; x = one-hot(21 atom types) ++ degree
                      code in the last 4 columns (libs/utils.py:253-259); y ~ N(0,1)
  C  counting-like  : G(n,p), n ~ U{10..30}, p = 0.3; x = [1]; targets = triangle
                      count trace(A^3)/6 (libs/utils.py:395-397)
  M  MNIST-75-like  : n = 75, k-NN graph on random 2-D points, 2 features U(0,1)
  R  sr25-like      : the 15 real strongly-regular graphs ship as a fixture
                      (tests/golden/data_sr25.npz); ``random_regular`` here is the
                      synthetic stand-in for benches.

Every generator returns a list of (x [n,f] float32, edge_index [2,e] int64, y).
edge_index is symmetric and in row-major ``np.where`` order like the reference's
dataset classes produce it.
"""
import numpy as np


def _edges_from_adj(A):
    r, c = np.where(A > 0)
    return np.vstack((r, c)).astype(np.int64)


def zinc_like_graph(rng, ntype=21, maxdeg=4):
    n = int(np.clip(np.rint(rng.normal(23.0, 4.0)), 9, 37))
    A = np.zeros((n, n), dtype=np.float32)
    deg = np.zeros(n, dtype=np.int64)
    for i in range(1, n):
        cand = np.flatnonzero(deg[:i] < maxdeg - (1 if i < n - 1 else 0))
        if cand.size == 0:
            cand = np.flatnonzero(deg[:i] < maxdeg)
        j = int(cand[rng.integers(cand.size)])
        A[i, j] = A[j, i] = 1
        deg[i] += 1
        deg[j] += 1
    extra = int(np.rint(0.08 * n))
    tries = 0
    while extra > 0 and tries < 50:
        tries += 1
        i, j = rng.integers(n, size=2)
        if i != j and A[i, j] == 0 and deg[i] < maxdeg and deg[j] < maxdeg:
            A[i, j] = A[j, i] = 1
            deg[i] += 1
            deg[j] += 1
            extra -= 1
    x = np.zeros((n, ntype + maxdeg), dtype=np.float32)
    types = rng.integers(ntype, size=n)
    x[np.arange(n), types] = 1
    x[np.arange(n), -deg] = 1          # deg 0 would alias column 0, as in the reference
    y = np.float32(rng.normal())
    return x, _edges_from_adj(A), y


def counting_like_graph(rng, nmin=10, nmax=30, p=0.3):
    n = int(rng.integers(nmin, nmax + 1))
    U = np.triu((rng.random((n, n)) < p).astype(np.float32), 1)
    A = U + U.T
    x = np.ones((n, 1), dtype=np.float32)
    A64 = A.astype(np.float64)
    tri = np.trace(A64 @ A64 @ A64) / 6.0
    return x, _edges_from_adj(A), np.float32(tri)


def mnist75_like_graph(rng, n=75, k=9):
    """k-NN (symmetrised) over random points: ~18 neighbours/node like MNISTSuperpixels."""
    pts = rng.random((n, 2))
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    nn = np.argsort(d2, axis=1)[:, :k]
    A = np.zeros((n, n), dtype=np.float32)
    A[np.repeat(np.arange(n), k), nn.ravel()] = 1
    A = np.maximum(A, A.T)
    x = rng.random((n, 2)).astype(np.float32)
    y = np.int64(rng.integers(10))
    return x, _edges_from_adj(A), y


def random_regular(rng, n=25, d=12):
    """d-regular circulant-with-shuffle graph (stand-in for the sr25 family)."""
    offs = rng.permutation(np.arange(1, n // 2 + 1))[:d // 2]
    A = np.zeros((n, n), dtype=np.float32)
    idx = np.arange(n)
    for o in offs:
        A[idx, (idx + o) % n] = 1
        A[(idx + o) % n, idx] = 1
    perm = rng.permutation(n)
    A = A[perm][:, perm]
    x = np.ones((n, 1), dtype=np.float32)
    return x, _edges_from_adj(A), np.float32(0)


_GEN = dict(zinc=zinc_like_graph, counting=counting_like_graph,
            mnist75=mnist75_like_graph, regular=random_regular)


def make_graphs(kind, count, seed=0, **kw):
    rng = np.random.default_rng(seed)
    gen = _GEN[kind]
    return [gen(rng, **kw) for _ in range(count)]
