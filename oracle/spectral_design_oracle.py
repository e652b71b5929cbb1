"""Oracle (CPU, numpy) for the support precompute ``SpectralDesign``.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Restates
/root/reference/libs/utils.py:546-610 (Alg.1 of the paper) including its dtype
walk: A, SP float32; (A+I), nL, eigh, the Gaussian filter products float64;
``M`` squared (not multiplied by A+I) ``recfield-1`` times (:566-573);
COO emitted in row-major ``np.where`` order (:608-610).  The PPGN tensors of
:613-624 are baseline-only and not produced.
"""
import numpy as np


def spectral_design(x, edge_index, recfield=1, dv=5, nfreq=5, adddegree=False,
                    laplacien=True, addadj=False, vmax=None):
    """x [n,f] float array, edge_index [2,e] int array.

    Returns dict(x [n,f(+1)] f32, edge_index2 [2,m] int64, edge_attr2 [m,S] f32,
                 lmax np.float32) -- the fields the hot path consumes.
    """
    x = np.asarray(x, dtype=np.float32)
    ei = np.asarray(edge_index, dtype=np.int64)
    n = x.shape[0]
    nsup = nfreq + 1 + (1 if addadj else 0)

    A = np.zeros((n, n), dtype=np.float32)                       # :558
    SP = np.zeros((nsup, n, n), dtype=np.float32)                # :559
    A[ei[0], ei[1]] = 1                                          # :560

    if adddegree:                                                # :562-563
        x = np.concatenate([x, A.sum(0)[:, None]], 1).astype(np.float32)

    if recfield == 0:                                            # :566-573
        M = A
    else:
        M = A + np.eye(n)
        for _ in range(1, recfield):
            M = M.dot(M)
    M = M > 0

    d = A.sum(axis=0)                                            # :576-582
    with np.errstate(divide='ignore', invalid='ignore'):
        dis = 1 / np.sqrt(d)
    dis[np.isinf(dis)] = 0
    dis[np.isnan(dis)] = 0
    D = np.diag(dis)
    nL = np.eye(n) - (A.dot(D)).T.dot(D)
    V, U = np.linalg.eigh(nL)                                    # :583
    V[V < 0] = 0
    lmax = V.max().astype(np.float32)                            # :586

    if not laplacien:                                            # :588-589
        V, U = np.linalg.eigh(A)

    top = V.max() if vmax is None else vmax                      # :592-596
    centers = np.linspace(V.min(), top, nfreq)

    for i in range(len(centers)):                                # :599-600
        SP[i] = M * (U.dot(np.diag(np.exp(-(dv * (V - centers[i]) ** 2))).dot(U.T)))
    SP[len(centers)] = np.eye(n)                                 # :602
    if addadj:                                                   # :604-605
        SP[len(centers) + 1] = A

    E = np.where(M > 0)                                          # :608-610
    edge_index2 = np.vstack((E[0], E[1])).astype(np.int64)
    edge_attr2 = np.ascontiguousarray(SP[:, E[0], E[1]].T).astype(np.float32)
    return dict(x=x, edge_index2=edge_index2, edge_attr2=edge_attr2, lmax=lmax)
