// GNNML1 block in one launch each way -- /root/reference/sr25.py:231-240, graph8c.py (the same class), mnist75.py:296-318,
// mutag.py:253-262:
//
//   a = fc_i1(x)   c = conv_i1(x)   f2 = fc_i2(x)   f3 = fc_i3(x)          conv_i1 = SpectConv(K = 1, selfconn = False): c = (A^T x) Wc + bc
//   mode 0 (SUM,      the scripts' setting):  out = act(a + c + f2 * f3)                                 [N, n]
//   mode 1 (CAT_PROD, concat = True):         out = [act(a) | act(c) | act(f2 * f3)]                     [N, n1 + n2 + n3]
//   mode 2 (CAT_FACT, mutag.py):              out = [act(a) | act(c) | act(f2) * act(f3)]
//
// Rounds 1-4 ran the block as three library Linears, one S = 1 SpectConv launch, a product, two sums / a concatenation and the
// activation: ~9 launches forward, ~25 backward, every intermediate through HBM.  Here: ONE launch forward; backward = one launch
// (dx, and the pre-activation gradients [da | dc | df2 | df3] + Q = A dc as arrays for the weight gradients) + gml_xty per weight.
//
// Arithmetic: exact fp32 (v_mfma_f32_16x16x4_f32 == an fmaf chain; VALU aggregation in the CSR's edge order = the reference's CPU
// scatter order).  One 16-row tile per wave, lane l = (r16 = l & 15, kq = l >> 4) owns row r16 and the features
// [kq FPL, kq FPL + FPL) of it (FP = 4 FPL = Fin rounded up to 16 / 64).  Every product is computed TRANSPOSED,
//     D^T[i = column][j = row] = sum_k  A[i][k] . B[k][j],      A = weight fragment (LDS, one dword per lane and MFMA, conflict-free),
//                                                                B = the lane's own register (x, the aggregate, a gradient)
// so that the lane (row, kq) receives columns 16 nb + 4 kq .. + 3 of ITS row: results of one MFMA chain are operands of the next
// without a shuffle, and rows move as 16-byte accesses.
#include "gml_common.h"

struct GmlG1Params {
    const int32_t* rowptr; const int32_t* col; const float* val;      // fwd: target-keyed CSR; bwd: source-keyed (rowptr_t, col_t, val_t)
    const float* x; int64_t ldx;
    const float* w1; const float* b1; const float* wc; const float* bc;
    const float* w2; const float* b2; const float* w3; const float* b3;
    float* out; int64_t ldo;                                             // fwd: written; bwd: the saved output (read)
    const float* gout; int64_t ldgo;
    float* dx; int64_t lddx; float* g4; int64_t ldg4; float* q; int64_t ldq;
    int64_t nrows; int32_t Fin, n1, n2, n3, mode, act, ntiles;
};

#define G1_NW 8
template <int FPL>
struct GmlG1Cfg {
    static constexpr int FP = 4 * FPL;
    // forward-form fragments [block][j < FPL][64 lanes]: lane (c = l & 15, k = l >> 4) holds W[16 nb + c][k FPL + j]
    __host__ __device__ static int fwd_floats(int nblk) { return nblk * FPL * 64; }
    // transposed-form fragments for dx: [fb][K block][4][64]: lane (f = l & 15, k = l >> 4) holds W[16 nb + 4 k + reg][16 fb + f]
    __host__ __device__ static int tr_floats(int nkb) { return (FP / 16) * nkb * 4 * 64; }
};

__device__ __forceinline__ float g1_act(float v, int act) { return act == 0 ? gml_tanh(v) : fmaxf(v, 0.f); }
// derivative of the activation from its OUTPUT value (tanh: 1 - y^2; relu: y > 0)
__device__ __forceinline__ float g1_dact_out(float y, int act) { return act == 0 ? fmaf(-y, y, 1.f) : (y > 0.f ? 1.f : 0.f); }

template <int FPL>
__device__ __forceinline__ void g1_fill_fwd(float* dst, const float* w, int n, int Fin, bool conv, int nb0, int nblk, int tid, int nt) {
    // blocks nb0 .. nb0 + nblk of one matrix; Linear weights are [n, Fin] row-major, the conv weight [Fin, n]
    for (int i = tid; i < nblk * FPL * 64; i += nt) {
        const int lane = i & 63, j = (i >> 6) % FPL, nb = (i >> 6) / FPL;
        const int c = 16 * nb + (lane & 15), f = (lane >> 4) * FPL + j;
        dst[(nb0 + nb) * FPL * 64 + j * 64 + lane] = (c < n && f < Fin) ? (conv ? w[(int64_t)f * n + c] : w[(int64_t)c * Fin + f]) : 0.f;
    }
}

template <int FPL>
__device__ __forceinline__ void g1_load_row(const float* x, int64_t ldx, int64_t row, bool valid, int Fin, int kq, float (&xr)[FPL]) {
    const float* xp = x + row * ldx + kq * FPL;
    const bool vec = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
#pragma unroll
    for (int j4 = 0; j4 < FPL / 4; ++j4) {
        const int f = kq * FPL + 4 * j4;
        if (valid && vec && f + 4 <= Fin) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(xp + 4 * j4);
            xr[4 * j4] = t.x; xr[4 * j4 + 1] = t.y; xr[4 * j4 + 2] = t.z; xr[4 * j4 + 3] = t.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) xr[4 * j4 + u] = (valid && f + u < Fin) ? xp[4 * j4 + u] : 0.f;
        }
    }
}

// one block of a transposed product: acc[reg] = column 16 nb + 4 kq + reg of the lane's row
template <int FPL>
__device__ __forceinline__ f32x4 g1_block(const float* wl, int blk, int lane, const float (&b)[FPL]) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wp = wl + blk * FPL * 64 + lane;
#pragma unroll
    for (int j = 0; j < FPL; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[j * 64], b[j], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ f32x4 g1_bias4(const float* b, int c0, int n) {
    f32x4 r;
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = (b && c0 + u < n) ? b[c0 + u] : 0.f;
    return r;
}

__device__ __forceinline__ void g1_store4(float* base, int64_t ld, int64_t row, int c0, int n, bool valid, f32x4 v) {
    if (!valid) return;
    float* p = base + row * ld + c0;
    if (c0 + 4 <= n && ld % 4 == 0 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) { *reinterpret_cast<f32x4*>(p) = v; return; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (c0 + u < n) p[u] = v[u];
}

__device__ __forceinline__ f32x4 g1_load4(const float* base, int64_t ld, int64_t row, int c0, int n, bool valid) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!valid) return v;
    const float* p = base + row * ld + c0;
    if (c0 + 4 <= n && ld % 4 == 0 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) return *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (c0 + u < n) v[u] = p[u];
    return v;
}

// ------------------------------------------------------------------------------------------------------------------ forward
template <int FPL>
__global__ __launch_bounds__(64 * G1_NW) void gml_k_gnnml1_fwd(const GmlG1Params p) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    const int nb1 = (p.n1 + 15) / 16, nb2 = (p.n2 + 15) / 16, nb3 = (p.n3 + 15) / 16;
    const int o1 = 0, oc = nb1, o2 = nb1 + nb2, o3 = nb1 + nb2 + nb3;            // block offsets of the four matrices in the image
    g1_fill_fwd<FPL>(wl, p.w1, p.n1, p.Fin, false, o1, nb1, tid, blockDim.x);
    g1_fill_fwd<FPL>(wl, p.wc, p.n2, p.Fin, true, oc, nb2, tid, blockDim.x);
    g1_fill_fwd<FPL>(wl, p.w2, p.n3, p.Fin, false, o2, nb3, tid, blockDim.x);
    g1_fill_fwd<FPL>(wl, p.w3, p.n3, p.Fin, false, o3, nb3, tid, blockDim.x);
    __syncthreads();
    const int cat_c = p.n1, cat_p = p.n1 + p.n2;                                    // column bases of the concatenated output
    for (int t = blockIdx.x * G1_NW + wave; t < p.ntiles; t += gridDim.x * G1_NW) {
        const int64_t row = (int64_t)t * 16 + r16;
        const bool valid = row < p.nrows;
        float xr[FPL], hr[FPL];
        g1_load_row<FPL>(p.x, p.ldx, row, valid, p.Fin, kq, xr);
#pragma unroll
        for (int j = 0; j < FPL; ++j) hr[j] = 0.f;
        const int e0 = valid ? p.rowptr[row] : 0, e1 = valid ? p.rowptr[row + 1] : 0;
        for (int e = e0; e < e1; ++e) {                                            // the reference's per-target summation order
            const int c = p.col[e];
            const float v = p.val ? p.val[e] : 1.f;
            float xn[FPL];
            g1_load_row<FPL>(p.x, p.ldx, c, true, p.Fin, kq, xn);
#pragma unroll
            for (int j = 0; j < FPL; ++j) hr[j] = fmaf(v, xn[j], hr[j]);
        }
        if (p.mode == 0) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (nb >= nb1) break;
                const int c0 = 16 * nb + 4 * kq;
                const f32x4 a = g1_block<FPL>(wl, o1 + nb, lane, xr), c = g1_block<FPL>(wl, oc + nb, lane, hr);
                const f32x4 f2 = g1_block<FPL>(wl, o2 + nb, lane, xr) + g1_bias4(p.b2, c0, p.n3);
                const f32x4 f3 = g1_block<FPL>(wl, o3 + nb, lane, xr) + g1_bias4(p.b3, c0, p.n3);
                const f32x4 v = (a + g1_bias4(p.b1, c0, p.n1)) + (c + g1_bias4(p.bc, c0, p.n2)) + f2 * f3;
                f32x4 y;
#pragma unroll
                for (int u = 0; u < 4; ++u) y[u] = g1_act(v[u], p.act);
                g1_store4(p.out, p.ldo, row, c0, p.n1, valid, y);
            }
        } else {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (nb < nb1) {
                    const int c0 = 16 * nb + 4 * kq;
                    const f32x4 v = g1_block<FPL>(wl, o1 + nb, lane, xr) + g1_bias4(p.b1, c0, p.n1);
                    f32x4 y;
#pragma unroll
                    for (int u = 0; u < 4; ++u) y[u] = g1_act(v[u], p.act);
                    g1_store4(p.out, p.ldo, row, c0, p.n1, valid, y);
                }
                if (nb < nb2) {
                    const int c0 = 16 * nb + 4 * kq;
                    const f32x4 v = g1_block<FPL>(wl, oc + nb, lane, hr) + g1_bias4(p.bc, c0, p.n2);
                    f32x4 y;
#pragma unroll
                    for (int u = 0; u < 4; ++u) y[u] = g1_act(v[u], p.act);
                    g1_store4(p.out + cat_c, p.ldo, row, c0, p.n2, valid, y);
                }
                if (nb < nb3) {
                    const int c0 = 16 * nb + 4 * kq;
                    const f32x4 f2 = g1_block<FPL>(wl, o2 + nb, lane, xr) + g1_bias4(p.b2, c0, p.n3);
                    const f32x4 f3 = g1_block<FPL>(wl, o3 + nb, lane, xr) + g1_bias4(p.b3, c0, p.n3);
                    f32x4 y;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        y[u] = p.mode == 1 ? g1_act(f2[u] * f3[u], p.act) : g1_act(f2[u], p.act) * g1_act(f3[u], p.act);
                    g1_store4(p.out + cat_p, p.ldo, row, c0, p.n3, valid, y);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ backward
// dx[r] = da[r] W1 + df2[r] W2 + df3[r] W3 + (sum_{e: src = r} val_e dc[dst_e]) Wc^T;   g4 = [da | dc | df2 | df3] (blocks of 16 nb
// columns, the SUM mode without the dc block: dc = da there) and q = A dc [N, 16 nb2] are written for the weight gradients:
//   dW1 = da^T x, dW2 = df2^T x, dW3 = df3^T x (gml_xty),  dWc = x^T q,  db = column sums of g4.
// Two launches (PH = 1, then PH = 2).  One kernel doing both had to form dc of a NEIGHBOUR row on the fly (two gathered 16-byte
// pieces per block and edge: gout and out) and kept 96 KB of weight images -- one workgroup per CU, 0.85 ms per launch on sr25's
// 13-entry rows, bound by the gathers at two waves per SIMD.  Phase 1 (forward-form W2 / W3 only: 32 KB) recomputes the factors and
// writes g4; phase 2 (transposed forms: 64 KB, two workgroups per CU) gathers dc from g4 -- half the gathered bytes --, re-reads its
// own row's blocks, writes q and forms dx.
template <int FPL, int PH>
__global__ __launch_bounds__(64 * G1_NW) void gml_k_gnnml1_bwd(const GmlG1Params p) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    using C = GmlG1Cfg<FPL>;
    constexpr int NFB = C::FP / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    const int nb1 = (p.n1 + 15) / 16, nb2 = (p.n2 + 15) / 16, nb3 = (p.n3 + 15) / 16;
    // image: forward-form W2, W3 (the factors are recomputed), then the transposed forms of W1, W2, W3, Wc
    float* wf = wl;                                           // [2 nb3][FPL][64]   (phase 1 only)
    float* wt1 = PH == 1 ? wl : wl;                           // [fb][nb1][4][64]   (phase 2 only: the image starts here)
    float* wt2 = wt1 + C::tr_floats(nb1);
    float* wt3 = wt2 + C::tr_floats(nb3);
    float* wtc = wt3 + C::tr_floats(nb3);                     // [fb][nb2][4][64]: lane (f, k) holds Wc[16 fb + f][16 nb + 4 k + reg]
    if constexpr (PH == 1) {
        g1_fill_fwd<FPL>(wf, p.w2, p.n3, p.Fin, false, 0, nb3, tid, blockDim.x);
        g1_fill_fwd<FPL>(wf, p.w3, p.n3, p.Fin, false, nb3, nb3, tid, blockDim.x);
    }
    auto fill_tr = [&](float* dst, const float* w, int n, int nkb, bool conv) {
        for (int i = tid; i < NFB * nkb * 4 * 64; i += blockDim.x) {
            const int ln = i & 63, reg = (i >> 6) & 3, nb = (i >> 8) % nkb, fb = (i >> 8) / nkb;
            const int c = 16 * nb + 4 * (ln >> 4) + reg, f = 16 * fb + (ln & 15);
            dst[i] = (c < n && f < p.Fin) ? (conv ? w[(int64_t)f * n + c] : w[(int64_t)c * p.Fin + f]) : 0.f;
        }
    };
    if (PH == 2 && p.dx) {
        fill_tr(wt1, p.w1, p.n1, nb1, false);
        fill_tr(wt2, p.w2, p.n3, nb3, false);
        fill_tr(wt3, p.w3, p.n3, nb3, false);
        fill_tr(wtc, p.wc, p.n2, nb2, true);
    }
    __syncthreads();
    const int sum = p.mode == 0;
    const int oc_c = sum ? 0 : p.n1, op_c = sum ? 0 : p.n1 + p.n2;               // column bases in out / gout
    const int ga_o = 0, gc_o = 16 * nb1, g2_o = sum ? 16 * nb1 : 16 * (nb1 + nb2), g3_o = g2_o + 16 * nb3;   // column bases in g4
    for (int t = blockIdx.x * G1_NW + wave; t < p.ntiles; t += gridDim.x * G1_NW) {
        const int64_t row = (int64_t)t * 16 + r16;
        const bool valid = row < p.nrows;
        f32x4 da[4], dc[4], d2[4], d3[4];
        if constexpr (PH == 2) {                             // the own row's blocks back from g4
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int c0 = 16 * nb + 4 * kq;
                da[nb] = (nb < nb1 && p.dx) ? g1_load4(p.g4 + ga_o, p.ldg4, row, c0, 16 * nb1, valid) : f32x4{0.f, 0.f, 0.f, 0.f};
                d2[nb] = (nb < nb3 && p.dx) ? g1_load4(p.g4 + g2_o, p.ldg4, row, c0, 16 * nb3, valid) : f32x4{0.f, 0.f, 0.f, 0.f};
                d3[nb] = (nb < nb3 && p.dx) ? g1_load4(p.g4 + g3_o, p.ldg4, row, c0, 16 * nb3, valid) : f32x4{0.f, 0.f, 0.f, 0.f};
                dc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
        float xr[FPL];
        g1_load_row<FPL>(p.x, p.ldx, row, valid, p.Fin, kq, xr);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            da[nb] = dc[nb] = d2[nb] = d3[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int c0 = 16 * nb + 4 * kq;
            f32x4 f2 = f32x4{0.f, 0.f, 0.f, 0.f}, f3 = f2;
            if (nb < nb3) {
                f2 = g1_block<FPL>(wf, nb, lane, xr) + g1_bias4(p.b2, c0, p.n3);
                f3 = g1_block<FPL>(wf, nb3 + nb, lane, xr) + g1_bias4(p.b3, c0, p.n3);
            }
            if (sum) {
                if (nb < nb1) {
                    const f32x4 go = g1_load4(p.gout, p.ldgo, row, c0, p.n1, valid), oo = g1_load4(p.out, p.ldo, row, c0, p.n1, valid);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float g = go[u] * g1_dact_out(oo[u], p.act);
                        da[nb][u] = g; dc[nb][u] = g; d2[nb][u] = g * f3[u]; d3[nb][u] = g * f2[u];
                    }
                }
            } else {
                if (nb < nb1) {
                    const f32x4 go = g1_load4(p.gout, p.ldgo, row, c0, p.n1, valid), oo = g1_load4(p.out, p.ldo, row, c0, p.n1, valid);
#pragma unroll
                    for (int u = 0; u < 4; ++u) da[nb][u] = go[u] * g1_dact_out(oo[u], p.act);
                }
                if (nb < nb2) {
                    const f32x4 go = g1_load4(p.gout + oc_c, p.ldgo, row, c0, p.n2, valid), oo = g1_load4(p.out + oc_c, p.ldo, row, c0, p.n2, valid);
#pragma unroll
                    for (int u = 0; u < 4; ++u) dc[nb][u] = go[u] * g1_dact_out(oo[u], p.act);
                }
                if (nb < nb3) {
                    const f32x4 go = g1_load4(p.gout + op_c, p.ldgo, row, c0, p.n3, valid), oo = g1_load4(p.out + op_c, p.ldo, row, c0, p.n3, valid);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (p.mode == 1) {
                            const float g = go[u] * g1_dact_out(oo[u], p.act);
                            d2[nb][u] = g * f3[u]; d3[nb][u] = g * f2[u];
                        } else {                                 // act(f2) act(f3): the factor's own activation and derivative
                            float a2, a3, e2, e3;
                            if (p.act == 0) { gml_tanh_d(f2[u], a2, e2); gml_tanh_d(f3[u], a3, e3); }
                            else { a2 = fmaxf(f2[u], 0.f); a3 = fmaxf(f3[u], 0.f); e2 = f2[u] > 0.f ? 1.f : 0.f; e3 = f3[u] > 0.f ? 1.f : 0.f; }
                            d2[nb][u] = go[u] * a3 * e2; d3[nb][u] = go[u] * a2 * e3;
                        }
                    }
                }
            }
            if (nb < nb1) g1_store4(p.g4 + ga_o, p.ldg4, row, c0, 16 * nb1, valid, da[nb]);
            if (!sum && nb < nb2) g1_store4(p.g4 + gc_o, p.ldg4, row, c0, 16 * nb2, valid, dc[nb]);
            if (nb < nb3) { g1_store4(p.g4 + g2_o, p.ldg4, row, c0, 16 * nb3, valid, d2[nb]); g1_store4(p.g4 + g3_o, p.ldg4, row, c0, 16 * nb3, valid, d3[nb]); }
        }
        continue;                                            // (phase 1 ends here)
        }
        // q[row] = sum over the row's OUT-edges (source-keyed view) of val * dc[destination], dc read from g4 (mode 0: its da block)
        const float* dcb = p.g4 + (sum ? ga_o : gc_o);
        f32x4 qa[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) qa[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int e0 = valid ? p.rowptr[row] : 0, e1 = valid ? p.rowptr[row + 1] : 0;
        for (int e = e0; e < e1; ++e) {
            const int64_t d = p.col[e];
            const float v = p.val ? p.val[e] : 1.f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (nb >= nb2) break;
                const int c0 = 16 * nb + 4 * kq;
                const f32x4 dn = g1_load4(dcb, p.ldg4, d, c0, 16 * nb2, true);
#pragma unroll
                for (int u = 0; u < 4; ++u) qa[nb][u] = fmaf(v, dn[u], qa[nb][u]);
            }
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            if (nb < nb2) g1_store4(p.q, p.ldq, row, 16 * nb + 4 * kq, 16 * nb2, valid, qa[nb]);
        if (p.dx) {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) {
                if (16 * fb >= p.Fin) break;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        if (nb < nb1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wt1[((fb * nb1 + nb) * 4 + reg) * 64 + lane], da[nb][reg], acc, 0, 0, 0);
                        if (nb < nb3) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wt2[((fb * nb3 + nb) * 4 + reg) * 64 + lane], d2[nb][reg], acc, 0, 0, 0);
                        if (nb < nb3) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wt3[((fb * nb3 + nb) * 4 + reg) * 64 + lane], d3[nb][reg], acc, 0, 0, 0);
                        if (nb < nb2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wtc[((fb * nb2 + nb) * 4 + reg) * 64 + lane], qa[nb][reg], acc, 0, 0, 0);
                    }
                }
                g1_store4(p.dx, p.lddx, row, 16 * fb + 4 * kq, p.Fin, valid, acc);
            }
        }
    }
}

static int g1_fpl(int Fin) { return Fin <= 16 ? 4 : (Fin <= 64 ? 16 : 0); }

static int g1_check(const GmlG1Params& p) {
    if (p.nrows < 0 || p.Fin <= 0 || p.n1 <= 0 || p.n2 <= 0 || p.n3 <= 0 || p.mode < 0 || p.mode > 2 || p.act < 0 || p.act > 1) return GML_E_BADARG;
    if (p.Fin > 64 || p.n1 > 64 || p.n2 > 64 || p.n3 > 64) return GML_E_UNSUPPORTED;
    if (p.mode == 0 && (p.n1 != p.n2 || p.n1 != p.n3)) return GML_E_BADARG;
    if (!p.rowptr || !p.col || !p.x || !p.w1 || !p.wc || !p.w2 || !p.w3 || !p.out) return GML_E_BADARG;
    return GML_OK;
}

// 1 when gml_gnnml1_fwd / _bwd serve these widths
extern "C" int gml_gnnml1_supported(int32_t Fin, int32_t n1, int32_t n2, int32_t n3, int32_t mode) {
    if (Fin <= 0 || n1 <= 0 || n2 <= 0 || n3 <= 0 || Fin > 64 || n1 > 64 || n2 > 64 || n3 > 64 || mode < 0 || mode > 2) return 0;
    return (mode != 0 || (n1 == n2 && n1 == n3)) ? 1 : 0;
}

extern "C" int gml_gnnml1_fwd(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t ldx, int64_t num_rows,
                              int32_t Fin, const float* w1, const float* b1, int32_t n1, const float* wc, const float* bc, int32_t n2,
                              const float* w2, const float* b2, const float* w3, const float* b3, int32_t n3, int32_t mode, int32_t act,
                              float* out, int64_t ldo, gml_stream_t stream) {
    GmlG1Params p = {};
    p.rowptr = rowptr; p.col = col; p.val = val; p.x = x; p.ldx = ldx; p.w1 = w1; p.b1 = b1; p.wc = wc; p.bc = bc; p.w2 = w2; p.b2 = b2;
    p.w3 = w3; p.b3 = b3; p.out = out; p.ldo = ldo; p.nrows = num_rows; p.Fin = Fin; p.n1 = n1; p.n2 = n2; p.n3 = n3; p.mode = mode; p.act = act;
    const int rc = g1_check(p);
    if (rc != GML_OK) return rc;
    if (ldx < Fin || ldo < (mode == 0 ? n1 : n1 + n2 + n3)) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    p.ntiles = (int)gml_cdiv(num_rows, 16);
    const int fpl = g1_fpl(Fin);
    const int nblk = (n1 + 15) / 16 + (n2 + 15) / 16 + 2 * ((n3 + 15) / 16);
    hipStream_t st = (hipStream_t)stream;
    int64_t grid = gml_cdiv(p.ntiles, G1_NW);
    if (grid > 2 * GML_NUM_CU) grid = 2 * GML_NUM_CU;
    if (fpl == 4) {
        const size_t lds = (size_t)GmlG1Cfg<4>::fwd_floats(nblk) * 4;
        GML_ALLOW_BIG_LDS(rc4, (&gml_k_gnnml1_fwd<4>), 160 * 1024)
        if (rc4 != hipSuccess) return (int)rc4;
        hipLaunchKernelGGL((gml_k_gnnml1_fwd<4>), dim3((unsigned)grid), dim3(64 * G1_NW), lds, st, p);
    } else {
        const size_t lds = (size_t)GmlG1Cfg<16>::fwd_floats(nblk) * 4;
        GML_ALLOW_BIG_LDS(rc16, (&gml_k_gnnml1_fwd<16>), 160 * 1024)
        if (rc16 != hipSuccess) return (int)rc16;
        if (lds > 64 * 1024 && grid > GML_NUM_CU) grid = GML_NUM_CU;
        hipLaunchKernelGGL((gml_k_gnnml1_fwd<16>), dim3((unsigned)grid), dim3(64 * G1_NW), lds, st, p);
    }
    return gml_launch_status();
}

// columns of g4: gml_gnnml1_g4_cols(n1, n2, n3, mode); q: [N, 16 ceil(n2 / 16)]
extern "C" int gml_gnnml1_g4_cols(int32_t n1, int32_t n2, int32_t n3, int32_t mode) {
    return 16 * ((n1 + 15) / 16 + (mode == 0 ? 0 : (n2 + 15) / 16) + 2 * ((n3 + 15) / 16));
}

extern "C" int gml_gnnml1_bwd(const int32_t* rowptr_t, const int32_t* col_t, const float* val_t, const float* x, int64_t ldx,
                              const float* out, int64_t ldo, const float* gout, int64_t ldgo, int64_t num_rows, int32_t Fin,
                              const float* w1, int32_t n1, const float* wc, int32_t n2, const float* w2, const float* b2, const float* w3,
                              const float* b3, int32_t n3, int32_t mode, int32_t act, float* dx, int64_t lddx, float* g4, int64_t ldg4,
                              float* q, int64_t ldq, gml_stream_t stream) {
    GmlG1Params p = {};
    p.rowptr = rowptr_t; p.col = col_t; p.val = val_t; p.x = x; p.ldx = ldx; p.w1 = w1; p.wc = wc; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3;
    p.out = const_cast<float*>(out); p.ldo = ldo; p.gout = gout; p.ldgo = ldgo; p.dx = dx; p.lddx = lddx; p.g4 = g4; p.ldg4 = ldg4; p.q = q; p.ldq = ldq;
    p.nrows = num_rows; p.Fin = Fin; p.n1 = n1; p.n2 = n2; p.n3 = n3; p.mode = mode; p.act = act;
    const int rc = g1_check(p);
    if (rc != GML_OK) return rc;
    const int C = mode == 0 ? n1 : n1 + n2 + n3;
    if (!gout || !g4 || !q || ldx < Fin || ldo < C || ldgo < C || (dx && lddx < Fin)) return GML_E_BADARG;
    if (ldg4 < gml_gnnml1_g4_cols(n1, n2, n3, mode) || ldg4 % 4 != 0 || ldq < 16 * ((n2 + 15) / 16) || ldq % 4 != 0) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    p.ntiles = (int)gml_cdiv(num_rows, 16);
    const int fpl = g1_fpl(Fin);
    const int nb1 = (n1 + 15) / 16, nb2 = (n2 + 15) / 16, nb3 = (n3 + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    int64_t grid = gml_cdiv(p.ntiles, G1_NW);
    if (grid > 2 * GML_NUM_CU) grid = 2 * GML_NUM_CU;
    if (fpl == 4) {
        const size_t lds1 = (size_t)GmlG1Cfg<4>::fwd_floats(2 * nb3) * 4, lds2 = dx ? (size_t)GmlG1Cfg<4>::tr_floats(nb1 + 2 * nb3 + nb2) * 4 : 0;
        if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return GML_E_UNSUPPORTED;
        GML_ALLOW_BIG_LDS(rc4a, (&gml_k_gnnml1_bwd<4, 1>), 160 * 1024)
        if (rc4a != hipSuccess) return (int)rc4a;
        GML_ALLOW_BIG_LDS(rc4, (&gml_k_gnnml1_bwd<4, 2>), 160 * 1024)
        if (rc4 != hipSuccess) return (int)rc4;
        hipLaunchKernelGGL((gml_k_gnnml1_bwd<4, 1>), dim3((unsigned)grid), dim3(64 * G1_NW), lds1, st, p);
        hipLaunchKernelGGL((gml_k_gnnml1_bwd<4, 2>), dim3((unsigned)grid), dim3(64 * G1_NW), lds2, st, p);
    } else {
        const size_t lds1 = (size_t)GmlG1Cfg<16>::fwd_floats(2 * nb3) * 4, lds2 = dx ? (size_t)GmlG1Cfg<16>::tr_floats(nb1 + 2 * nb3 + nb2) * 4 : 0;
        if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return GML_E_UNSUPPORTED;
        GML_ALLOW_BIG_LDS(rc16a, (&gml_k_gnnml1_bwd<16, 1>), 160 * 1024)
        if (rc16a != hipSuccess) return (int)rc16a;
        GML_ALLOW_BIG_LDS(rc16, (&gml_k_gnnml1_bwd<16, 2>), 160 * 1024)
        if (rc16 != hipSuccess) return (int)rc16;
        hipLaunchKernelGGL((gml_k_gnnml1_bwd<16, 1>), dim3((unsigned)grid), dim3(64 * G1_NW), lds1, st, p);
        hipLaunchKernelGGL((gml_k_gnnml1_bwd<16, 2>), dim3((unsigned)grid), dim3(64 * G1_NW), lds2, st, p);
    }
    return gml_launch_status();
}

// ------------------------------------------------------------------------------------------------------- weight gradients
// dW1 = da^T x, dW2 = df2^T x, dW3 = df3^T x ([n, Fin]),  dWc = x^T q ([Fin, n2]),  column sums of g4 (the bias gradients): ONE pass
// over the rows.  Contraction over ROWS on the f32 matrix instruction: A[i][k = row 4 t + (l >> 4)], B[k][j] are dword loads of 64-byte
// row segments (lane l & 15 = column inside a 16-wide block); a "group" = one 16-column block of g4 against all x blocks (type 0) or
// one x block against all q blocks (dWc); wave w of a workgroup owns the g4 blocks w and w + 8 (<= 16 blocks: the 64-wide concat form),
// walks the rows of the workgroup's chunk ONCE for both (the x fragments of a step loaded once, 8 accumulator tiles), and -- waves
// 0 .. 3 -- a second time for its x block of dWc.  One partial per workgroup, laid out as
// the flat result [dW1 | dW2 | dW3 | dWc | sums], folded in workgroup order by gml_fold_many (deterministic).
struct GmlG1DwParams {
    const float* x; int64_t ldx; const float* g4; int64_t ldg4; const float* q; int64_t ldq;
    int64_t nrows; int32_t Fin, n1, n2, n3, mode;
    float* part; int64_t nflat; int32_t rows_per_wg;
};

__global__ __launch_bounds__(64 * G1_NW) void gml_k_gnnml1_dw(const GmlG1DwParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, k4 = lane >> 4;
    const int nb1 = (p.n1 + 15) / 16, nb2 = (p.n2 + 15) / 16, nb3 = (p.n3 + 15) / 16, nfb = (p.Fin + 15) / 16;
    const int ngb = nb1 + (p.mode == 0 ? 0 : nb2) + 2 * nb3;          // 16-column blocks of g4 (<= 16)
    const int64_t r_begin = (int64_t)blockIdx.x * p.rows_per_wg;
    const int64_t r_end = min(r_begin + (int64_t)p.rows_per_wg, p.nrows);
    float* out = p.part + (int64_t)blockIdx.x * p.nflat;
    // flat offsets: dW1 [n1, Fin], dW2 [n3, Fin], dW3 [n3, Fin], dWc [Fin, n2], sums [16 ngb]
    const int64_t o_w1 = 0, o_w2 = (int64_t)p.n1 * p.Fin, o_w3 = o_w2 + (int64_t)p.n3 * p.Fin, o_wc = o_w3 + (int64_t)p.n3 * p.Fin,
                  o_s = o_wc + (int64_t)p.Fin * p.n2;
    constexpr int U = 4;                                               // K steps (4 rows each) of loads in flight in front of their MFMAs
    // ---- pass 1: the wave's g4 blocks (wave, wave + 8: <= 2 of <= 16) against ALL x blocks, the x fragments loaded once per step
    {
        const int na = (wave < ngb ? 1 : 0) + (wave + G1_NW < ngb ? 1 : 0);
        f32x4 acc[2][4];
        float bsum[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[i][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (na > 0) {
            for (int64_t r = r_begin; r < r_end; r += 4 * U) {
                float av[U][2], bv[U][4];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t row = r + 4 * u + k4;
                    const bool rv = row < r_end;
#pragma unroll
                    for (int i = 0; i < 2; ++i) av[u][i] = (i < na && rv) ? p.g4[row * p.ldg4 + 16 * (wave + G1_NW * i) + c16] : 0.f;
#pragma unroll
                    for (int b = 0; b < 4; ++b) bv[u][b] = (b < nfb && rv && 16 * b + c16 < p.Fin) ? p.x[row * p.ldx + 16 * b + c16] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (i >= na) break;
                        bsum[i] += av[u][i];
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            if (b < nfb) acc[i][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[u][b], acc[i][b], 0, 0, 0);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i >= na) break;
            const int grp = wave + G1_NW * i;
            // which matrix does g4 block `grp` belong to: [da nb1 | dc nb2 (modes 1, 2) | df2 nb3 | df3 nb3]
            int m, nb;
            if (grp < nb1) { m = 0; nb = grp; }
            else if (p.mode != 0 && grp < nb1 + nb2) { m = 3; nb = grp - nb1; }
            else { const int g2 = grp - nb1 - (p.mode == 0 ? 0 : nb2); m = g2 < nb3 ? 1 : 2; nb = g2 < nb3 ? g2 : g2 - nb3; }
            if (m != 3) {                                             // (the dc block has no Linear weight: its sums only)
                const int n = m == 0 ? p.n1 : p.n3;
                float* w = out + (m == 0 ? o_w1 : (m == 1 ? o_w2 : o_w3));
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b >= nfb) break;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {               // D[i = g4 column 4 k4 + reg][j = x column c16]
                        const int c = 16 * nb + 4 * k4 + reg, f = 16 * b + c16;
                        if (c < n && f < p.Fin) w[(int64_t)c * p.Fin + f] = acc[i][b][reg];
                    }
                }
            }
            float t = bsum[i];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (k4 == 0) out[o_s + 16 * grp + c16] = t;
        }
    }
    // ---- pass 2: dWc = x^T q: x block `wave` (< nfb <= 4) against all q blocks
    if (wave < nfb) {
        const int fb = wave, acols = min(16, p.Fin - 16 * fb);
        f32x4 acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int64_t r = r_begin; r < r_end; r += 4 * U) {
            float av[U], bv[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = r + 4 * u + k4;
                const bool rv = row < r_end;
                av[u] = (rv && c16 < acols) ? p.x[row * p.ldx + 16 * fb + c16] : 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) bv[u][b] = (b < nb2 && rv) ? p.q[row * p.ldq + 16 * b + c16] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (b < nb2) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][b], acc[b], 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b >= nb2) break;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int f = 16 * fb + 4 * k4 + reg, c = 16 * b + c16;
                if (f < p.Fin && c < p.n2) out[o_wc + (int64_t)f * p.n2 + c] = acc[b][reg];
            }
        }
    }
}

// floats of the flat result [dW1 (n1 x Fin) | dW2 (n3 x Fin) | dW3 (n3 x Fin) | dWc (Fin x n2) | column sums of g4 (gml_gnnml1_g4_cols)]
extern "C" int64_t gml_gnnml1_dw_floats(int32_t Fin, int32_t n1, int32_t n2, int32_t n3, int32_t mode) {
    return (int64_t)n1 * Fin + 2 * (int64_t)n3 * Fin + (int64_t)Fin * n2 + gml_gnnml1_g4_cols(n1, n2, n3, mode);
}

static int g1_dw_grid(int64_t n) {
    int64_t g = gml_cdiv(n, 128);
    if (g > 2 * GML_NUM_CU) g = 2 * GML_NUM_CU;
    return g < 1 ? 1 : (int)g;
}

extern "C" size_t gml_gnnml1_dw_workspace_bytes(int64_t num_rows, int32_t Fin, int32_t n1, int32_t n2, int32_t n3, int32_t mode) {
    return (size_t)g1_dw_grid(num_rows) * (size_t)gml_gnnml1_dw_floats(Fin, n1, n2, n3, mode) * sizeof(float);
}

extern "C" int gml_gnnml1_dw(const float* x, int64_t ldx, const float* g4, int64_t ldg4, const float* q, int64_t ldq, int64_t num_rows,
                             int32_t Fin, int32_t n1, int32_t n2, int32_t n3, int32_t mode, float* out_flat, void* ws, size_t ws_bytes,
                             gml_stream_t stream) {
    if (!gml_gnnml1_supported(Fin, n1, n2, n3, mode)) return GML_E_UNSUPPORTED;
    if (num_rows < 0 || !x || !g4 || !q || !out_flat || ldx < Fin || ldg4 < gml_gnnml1_g4_cols(n1, n2, n3, mode) || ldq < 16 * ((n2 + 15) / 16))
        return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nflat = gml_gnnml1_dw_floats(Fin, n1, n2, n3, mode);
    if (num_rows == 0) { gml_zero_async(out_flat, sizeof(float) * nflat, st); return gml_launch_status(); }
    if (!ws || ws_bytes < gml_gnnml1_dw_workspace_bytes(num_rows, Fin, n1, n2, n3, mode)) return GML_E_WORKSPACE;
    GmlG1DwParams p;
    p.x = x; p.ldx = ldx; p.g4 = g4; p.ldg4 = ldg4; p.q = q; p.ldq = ldq; p.nrows = num_rows; p.Fin = Fin; p.n1 = n1; p.n2 = n2; p.n3 = n3;
    p.mode = mode; p.part = (float*)ws; p.nflat = nflat;
    const int grid = g1_dw_grid(num_rows);
    p.rows_per_wg = (int)((gml_cdiv(num_rows, grid) + 3) / 4 * 4);
    gml_zero_async(ws, (size_t)grid * nflat * sizeof(float), st);       // (elements outside the written blocks: the dc block has no weight)
    hipLaunchKernelGGL(gml_k_gnnml1_dw, dim3((unsigned)grid), dim3(64 * G1_NW), 0, st, p);
    int rc = gml_launch_status();
    if (rc != GML_OK) return rc;
    gml_fold_job job = {};
    job.partial = (const float*)ws; job.nparts = grid; job.n = nflat; job.dst[0] = out_flat; job.ndst[0] = nflat;
    return gml_fold_many(&job, 1, stream);
}
