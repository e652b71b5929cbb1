// Support precompute (Alg.1, /root/reference/libs/utils.py:546-610 `SpectralDesign.__call__`) for a whole batch on the
// device: the step immediately before the layer path.  One workgroup per graph, everything of a
// graph (adjacency, mask, Laplacian, eigenvectors) lives in LDS:
//
//   A[u,v] = 1 per edge                         M = A (recfield 0) or (A+I) squared recfield-1 times, > 0   (:566-573)
//   d = column sums, nL = I - D^-1/2 A^T D^-1/2 (:576-582), symmetric eigenproblem of nL in float64 (:583) by cyclic
//   two-sided Jacobi with the round-robin pairing: the n/2 rotations of a round touch disjoint rows / columns, so a
//   round is two barrier-separated sweeps over LDS (columns of L and U, then rows of L);
//   V[V<0] = 0, lmax = max V (:584-586); laplacien=False: eigenproblem of A instead (:588-589);
//   centers = linspace(min V, vmax or max V, nfreq), support_i = M .* (U exp(-dv (V - c_i)^2) U^T), support_nfreq = I,
//   [addadj: A]  (:592-605);   COO of M in row-major order with the support values per edge (:608-610).
//
// Two launches: gml_spectral_count (mask only -> #edges of the mask per graph) and, after an exclusive scan by the
// caller, gml_spectral_design (everything, written at the graph's offset).  Supports are invariant to the choice of
// eigenvectors (sign, rotation inside a degenerate eigenspace), so they agree with a LAPACK-based evaluation to
// float64 roundoff; the mask and its order are integer work (bit-exact).
#include "gml_common.h"

#define SPD_NMAX 80          /* nodes per graph (LDS: 2 n^2 doubles + n^2 floats + 2 n^2 bytes) */
#define SPD_FMAX 16          /* band-pass supports */
#define SPD_THREADS 256

struct GmlSpectralParams {
    const int32_t* node_ptr;     // [B+1]
    const int32_t* edge_ptr;     // [B+1] edges of graph b: [edge_ptr[b], edge_ptr[b+1]) of edge_index
    const int64_t* edge_index;   // [2, e_total], global node ids
    int64_t e_total;
    int32_t recfield, nfreq, addadj, laplacien, has_vmax;
    double dv, vmax;
    const int64_t* out_ptr;      // [B+1] exclusive scan of the mask sizes
    int64_t* edge_index2;        // [2, m_total]
    float* edge_attr2;           // [m_total, S]
    float* lmax;                 // [B]
    int32_t* nnz;                // count kernel: [B]
    int64_t m_total;
};

// adjacency (float, as the reference's A) and mask into LDS; returns with `M` pointing at the final mask buffer
__device__ __forceinline__ unsigned char* spd_build_mask(const GmlSpectralParams& p, int nb, int n, int eb, int ee, float* Af,
                                                         unsigned char* M0, unsigned char* M1) {
    const int tid = threadIdx.x;
    for (int i = tid; i < n * n; i += SPD_THREADS) Af[i] = 0.f;
    __syncthreads();
    for (int k = eb + tid; k < ee; k += SPD_THREADS) {
        const int u = (int)(p.edge_index[k] - nb), v = (int)(p.edge_index[p.e_total + k] - nb);
        if (u >= 0 && u < n && v >= 0 && v < n) Af[u * n + v] = 1.f;
    }
    __syncthreads();
    for (int i = tid; i < n * n; i += SPD_THREADS) {
        const int r = i / n, c = i - r * n;
        M0[i] = (Af[i] > 0.f || (p.recfield > 0 && r == c)) ? 1 : 0;
    }
    __syncthreads();
    unsigned char *src = M0, *dst = M1;
    for (int it = 1; it < p.recfield; ++it) {                 // M <- M M (boolean): 2^(recfield-1) hops
        for (int i = tid; i < n * n; i += SPD_THREADS) {
            const int r = i / n, c = i - r * n;
            unsigned char a = 0;
            for (int k = 0; k < n; ++k) a |= src[r * n + k] & src[k * n + c];
            dst[i] = a;
        }
        __syncthreads();
        unsigned char* t = src; src = dst; dst = t;
    }
    return src;
}

__global__ __launch_bounds__(SPD_THREADS) void gml_k_spectral_count(const GmlSpectralParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int cnt[SPD_THREADS / 64];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int nb = p.node_ptr[g], n = p.node_ptr[g + 1] - nb;
    float* Af = reinterpret_cast<float*>(lds_raw);
    unsigned char* M0 = reinterpret_cast<unsigned char*>(Af + n * n);
    unsigned char* M1 = M0 + n * n;
    const unsigned char* M = spd_build_mask(p, nb, n, p.edge_ptr[g], p.edge_ptr[g + 1], Af, M0, M1);
    int c = 0;
    for (int i = tid; i < n * n; i += SPD_THREADS) c += M[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    if ((tid & 63) == 0) cnt[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) p.nnz[g] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}

// cyclic two-sided Jacobi on the symmetric L [n,n] (LDS, float64); U accumulates the eigenvectors (columns)
__device__ void spd_jacobi(double* L, double* U, double* cs, double* red, int n) {
    const int tid = threadIdx.x;
    const int m = (n + 1) & ~1;                               // players of the round-robin (a bye when n is odd)
    const int np = m / 2;
    for (int i = tid; i < n * n; i += SPD_THREADS) U[i] = ((i / n) == (i % n)) ? 1.0 : 0.0;
    __syncthreads();
    for (int sweep = 0; sweep < 40; ++sweep) {
        // off-diagonal mass vs diagonal mass
        double off = 0.0, dia = 0.0;
        for (int i = tid; i < n * n; i += SPD_THREADS) {
            const int r = i / n, c = i - r * n;
            const double v = L[i] * L[i];
            if (r == c) dia += v; else off += v;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { off += __shfl_xor(off, o); dia += __shfl_xor(dia, o); }
        if ((tid & 63) == 0) { red[2 * (tid >> 6)] = off; red[2 * (tid >> 6) + 1] = dia; }
        __syncthreads();
        off = red[0] + red[2] + red[4] + red[6];
        dia = red[1] + red[3] + red[5] + red[7];
        __syncthreads();
        if (off <= 1e-30 * dia || off == 0.0) break;
        for (int r = 0; r < m - 1; ++r) {
            // rotation angles of the np disjoint pairs of this round
            if (tid < np) {
                int pp, qq;
                if (tid == 0) { pp = m - 1; qq = r; }
                else { pp = (r + tid) % (m - 1); qq = (r - tid + (m - 1)) % (m - 1); }
                double c = 1.0, s = 0.0;
                if (pp < n && qq < n) {
                    const double apq = L[pp * n + qq];
                    if (fabs(apq) > 1e-300) {
                        const double tau = (L[qq * n + qq] - L[pp * n + pp]) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        s = t * c;
                    }
                }
                cs[2 * tid] = c;
                cs[2 * tid + 1] = s;
            }
            __syncthreads();
            // columns p, q of L and U
            for (int it = tid; it < np * n; it += SPD_THREADS) {
                const int k = it / n, i = it - k * n;
                int pp, qq;
                if (k == 0) { pp = m - 1; qq = r; }
                else { pp = (r + k) % (m - 1); qq = (r - k + (m - 1)) % (m - 1); }
                if (pp >= n || qq >= n) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                const double lp = L[i * n + pp], lq = L[i * n + qq];
                L[i * n + pp] = c * lp - s * lq;
                L[i * n + qq] = s * lp + c * lq;
                const double up = U[i * n + pp], uq = U[i * n + qq];
                U[i * n + pp] = c * up - s * uq;
                U[i * n + qq] = s * up + c * uq;
            }
            __syncthreads();
            // rows p, q of L
            for (int it = tid; it < np * n; it += SPD_THREADS) {
                const int k = it / n, j = it - k * n;
                int pp, qq;
                if (k == 0) { pp = m - 1; qq = r; }
                else { pp = (r + k) % (m - 1); qq = (r - k + (m - 1)) % (m - 1); }
                if (pp >= n || qq >= n) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                const double lp = L[pp * n + j], lq = L[qq * n + j];
                L[pp * n + j] = c * lp - s * lq;
                L[qq * n + j] = s * lp + c * lq;
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(SPD_THREADS) void gml_k_spectral_design(const GmlSpectralParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ double red[8];
    __shared__ double cs[SPD_NMAX + 2];
    __shared__ double V[SPD_NMAX];
    __shared__ double dis[SPD_NMAX];
    __shared__ int rowoff[SPD_NMAX + 1];
    __shared__ double ctr[SPD_FMAX];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int nb = p.node_ptr[g], n = p.node_ptr[g + 1] - nb;
    if (n <= 0) return;
    double* L = reinterpret_cast<double*>(lds_raw);
    double* U = L + n * n;
    double* wgt = U + n * n;                                   // [nfreq][n]
    float* Af = reinterpret_cast<float*>(wgt + p.nfreq * n);
    unsigned char* M0 = reinterpret_cast<unsigned char*>(Af + n * n);
    unsigned char* M1 = M0 + n * n;
    unsigned char* M = spd_build_mask(p, nb, n, p.edge_ptr[g], p.edge_ptr[g + 1], Af, M0, M1);
    unsigned char* rank = (M == M0) ? M1 : M0;                 // column rank of every mask entry inside its row

    // d = column sums (float32 like the reference), 1/sqrt(d) with inf/nan -> 0
    if (tid < n) {
        float d = 0.f;
        for (int i = 0; i < n; ++i) d += Af[i * n + tid];
        float v = 1.f / sqrtf(d);
        if (isinf(v) || isnan(v)) v = 0.f;
        dis[tid] = (double)v;
    }
    __syncthreads();
    // nL = I - (A D)^T D: nL[j][i] = delta - A[i][j] dis[i] dis[j] (products formed in float32, :580-582); the symmetric
    // solver reads the lower triangle (numpy.linalg.eigh UPLO='L')
    for (int i = tid; i < n * n; i += SPD_THREADS) {
        const int r = i / n, c = i - r * n;
        const int hi = max(r, c), lo = min(r, c);              // entry (hi, lo) of nL = delta - A[lo][hi] dis[lo] dis[hi]
        const float t = (Af[lo * n + hi] * (float)dis[hi]) * (float)dis[lo];
        L[i] = ((r == c) ? 1.0 : 0.0) - (double)t;
    }
    __syncthreads();
    spd_jacobi(L, U, cs, red, n);
    if (tid < n) V[tid] = fmax(L[tid * n + tid], 0.0);
    __syncthreads();
    if (tid == 0) {
        double mx = V[0];
        for (int i = 1; i < n; ++i) mx = fmax(mx, V[i]);
        p.lmax[g] = (float)mx;
    }
    __syncthreads();
    if (!p.laplacien) {                                        // spectrum of A instead (lower triangle)
        for (int i = tid; i < n * n; i += SPD_THREADS) {
            const int r = i / n, c = i - r * n;
            L[i] = (double)Af[max(r, c) * n + min(r, c)];
        }
        __syncthreads();
        spd_jacobi(L, U, cs, red, n);
        if (tid < n) V[tid] = L[tid * n + tid];
        __syncthreads();
    }
    if (tid == 0) {
        double mn = V[0], mx = V[0];
        for (int i = 1; i < n; ++i) { mn = fmin(mn, V[i]); mx = fmax(mx, V[i]); }
        const double top = p.has_vmax ? p.vmax : mx;
        // numpy.linspace(mn, top, nfreq): start + i * step, last point = stop
        const int nf = p.nfreq;
        const double step = nf > 1 ? (top - mn) / (double)(nf - 1) : 0.0;
        for (int i = 0; i < nf; ++i) ctr[i] = (i == nf - 1 && nf > 1) ? top : mn + (double)i * step;
    }
    __syncthreads();
    for (int it = tid; it < p.nfreq * n; it += SPD_THREADS) {
        const int i = it / n, k = it - i * n;
        const double dlt = V[k] - ctr[i];
        wgt[it] = exp(-(p.dv * dlt * dlt));
    }
    // row-major positions of the mask entries
    if (tid < n) {
        int c = 0;
        for (int j = 0; j < n; ++j) { rank[tid * n + j] = (unsigned char)c; c += M[tid * n + j]; }
        rowoff[tid + 1] = c;
    }
    __syncthreads();
    if (tid == 0) {
        rowoff[0] = 0;
        for (int r = 0; r < n; ++r) rowoff[r + 1] += rowoff[r];
    }
    __syncthreads();
    const int S = p.nfreq + 1 + (p.addadj ? 1 : 0);
    const int64_t base = p.out_ptr[g];
    for (int i = tid; i < n * n; i += SPD_THREADS) {
        if (!M[i]) continue;
        const int r = i / n, c = i - r * n;
        const int64_t pos = base + rowoff[r] + rank[i];
        p.edge_index2[pos] = (int64_t)(nb + r);
        p.edge_index2[p.m_total + pos] = (int64_t)(nb + c);
        float* out = p.edge_attr2 + pos * S;
        for (int f = 0; f < p.nfreq; ++f) {
            double a = 0.0;
            const double* w = wgt + f * n;
            for (int k = 0; k < n; ++k) a += U[r * n + k] * w[k] * U[c * n + k];
            out[f] = (float)a;
        }
        out[p.nfreq] = (r == c) ? 1.f : 0.f;
        if (p.addadj) out[p.nfreq + 1] = Af[i];
    }
}

static size_t spd_lds_bytes(int nmax, int nfreq, bool full) {
    const size_t nn = (size_t)nmax * nmax;
    size_t b = nn * sizeof(float) + 2 * nn;                    // Af, M0, M1
    if (full) b += 2 * nn * sizeof(double) + (size_t)nfreq * nmax * sizeof(double);
    return (b + 15) & ~(size_t)15;
}

extern "C" int gml_spectral_count(const int32_t* node_ptr, const int32_t* edge_ptr, const int64_t* edge_index,
                                  int64_t e_total, int64_t num_graphs, int32_t max_nodes, int32_t recfield,
                                  int32_t* nnz, gml_stream_t stream) {
    if (num_graphs < 0 || e_total < 0 || recfield < 0 || max_nodes < 0) return GML_E_BADARG;
    if (num_graphs == 0) return GML_OK;
    if (!node_ptr || !edge_ptr || !nnz || (e_total > 0 && !edge_index)) return GML_E_BADARG;
    if (max_nodes > SPD_NMAX) return GML_E_UNSUPPORTED;
    GmlSpectralParams p = {};
    p.node_ptr = node_ptr; p.edge_ptr = edge_ptr; p.edge_index = edge_index; p.e_total = e_total;
    p.recfield = recfield; p.nnz = nnz;
    hipLaunchKernelGGL(gml_k_spectral_count, dim3((unsigned)num_graphs), dim3(SPD_THREADS),
                       spd_lds_bytes(max_nodes, 0, false), (hipStream_t)stream, p);
    return gml_launch_status();
}

extern "C" int gml_spectral_design(const int32_t* node_ptr, const int32_t* edge_ptr, const int64_t* edge_index,
                                   int64_t e_total, int64_t num_graphs, int32_t max_nodes, int32_t recfield,
                                   int32_t nfreq, double dv, int32_t has_vmax, double vmax, int32_t laplacien,
                                   int32_t addadj, const int64_t* out_ptr, int64_t m_total, int64_t* edge_index2,
                                   float* edge_attr2, float* lmax, gml_stream_t stream) {
    if (num_graphs < 0 || e_total < 0 || recfield < 0 || nfreq <= 0 || m_total < 0 || max_nodes < 0) return GML_E_BADARG;
    if (num_graphs == 0) return GML_OK;
    if (!node_ptr || !edge_ptr || !out_ptr || !lmax || (e_total > 0 && !edge_index) ||
        (m_total > 0 && (!edge_index2 || !edge_attr2)))
        return GML_E_BADARG;
    if (max_nodes > SPD_NMAX || nfreq > SPD_FMAX) return GML_E_UNSUPPORTED;
    GmlSpectralParams p = {};
    p.node_ptr = node_ptr; p.edge_ptr = edge_ptr; p.edge_index = edge_index; p.e_total = e_total;
    p.recfield = recfield; p.nfreq = nfreq; p.addadj = addadj; p.laplacien = laplacien; p.has_vmax = has_vmax;
    p.dv = dv; p.vmax = vmax; p.out_ptr = out_ptr; p.edge_index2 = edge_index2; p.edge_attr2 = edge_attr2;
    p.lmax = lmax; p.m_total = m_total;
    GML_ALLOW_BIG_LDS(arc, (&gml_k_spectral_design), 156 * 1024)
    if (arc != hipSuccess) return (int)arc;
    hipLaunchKernelGGL(gml_k_spectral_design, dim3((unsigned)num_graphs), dim3(SPD_THREADS),
                       spd_lds_bytes(max_nodes, nfreq, true), (hipStream_t)stream, p);
    return gml_launch_status();
}
