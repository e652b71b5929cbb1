// Fused backward of the multi-support spectral convolution for gfx950 -- ONE launch produces
//
//   dX[r, :]      = sum_s ( sum_{e out of r} val[e, s] * G[dst(e), :] ) @ W[s]^T
//   dval[e, s]    = < X[src(e), :] @ W[s] , G[dst(e), :] >
//   dW[s]         = sum_r X[r, :]^T ( sum_{e out of r} val[e, s] * G[dst(e), :] )
//
// i.e. the autograd of /root/reference/libs/spect_conv.py:76-80 (gather backward, message() backward
// w.r.t. x_j and norm, scatter backward, matmul backward) without materialising any [N, S*F] tensor.
// G = gradient at the conv output (after the relu mask).  Everything is organised by SOURCE rows
// (the CSR passed in is keyed by source, values in that order), so nothing is scattered: no atomics,
// bitwise deterministic.
//
// Workgroup = 4 waves = one group of 64 source rows at a time (16 per wave).  Per group:
//   stage   CSR slice, value rows, the window of G rows the group's targets fall into, the group's own
//           X rows -> LDS (coalesced);
//   Z       Z[r][s][o] = X[r] W[s]        v_mfma_f32_16x16x4_f32, computed transposed so lane
//                                         (r = l&15, kq = l>>4) holds o = ob*16 + 4*kq + reg;
//   edges   P[r][s][o] += val[e,s] G[dst][o];   d[s] = <Z[r][s][:], G[dst][:]>  -> quad reduce -> dval
//   dX      P is already the A fragment of the (s,o)-contraction: 16x16x4 MFMAs against W^T from LDS
//   dW      contraction over ROWS: P goes through a small LDS exchange buffer (one SE-support slab at a
//           time); wave w owns the (s, fb, ob) output blocks with index = w (mod 4) and keeps their
//           accumulators in registers across all its groups; one partial per workgroup at the end.
#pragma once
#include "gml_common.h"

struct GmlBwdParams {
    const int32_t* rowptr;
    const int32_t* col;
    const int32_t* ginfo;
    const float* val;
    const float* x;
    int64_t ldx;
    const float* g;
    int64_t ldg;
    const float* w;
    float* dx;
    int64_t lddx;
    float* dval;
    float* dw_partial;
    int64_t nrows;
    int32_t S, Fin, Fout;
    uint32_t flags;
    int32_t ngroups, groups_per_wg;
    int32_t ecap, xcap;      // LDS capacities (edges per group, G-window rows), multiples of 4
    int32_t xvec, gvec;      // x / g rows may be read as aligned float4
    int32_t dxvec;           // dx rows may be read / written as aligned float4 (bwd3)
    // bwd3, optional: dx = conv part + dz wmix (dz [N, 4] contiguous, wmix [nmix <= 4, Fin]): the ML3Layer Hadamard branch's
    // share of dx handed over as its 4 pre-activation gradients per row instead of a written and re-read [N, Fin] array
    const float* dz;
    const float* wmix;
    int32_t nmix;
    const float* wmix2;      // optional second array: rows nmix1 .. nmix - 1 of wmix come from here (fc11.weight, fc12.weight as they are: no concatenation)
    int32_t nmix1;           // rows taken from wmix (= nmix when wmix2 is NULL)
    // bwd3 DZ form, optional: the first relu_cols features of x are relu outputs of the layer below (x = [relu(conv) | ...], an
    // ML3Layer feeding an ML3Layer): dx[:, f < relu_cols] is written already multiplied by (x[:, f] > 0) -- it IS the gradient at
    // that layer's conv output, so its output-stage backward needs neither its saved output nor a second [N, C] array
    int32_t relu_cols;
    // bwd3 HAD form (gml_spectconv_bwd_had; ZINC's 30 + 2 layers): the whole output stage of the layer runs inside the conv backward.
    // g = the pre-masked gradient at the layer output [N, Fout + 2] (columns Fout, Fout + 1: the Hadamard branch's), wmix / wmix2 =
    // fc11.weight / fc12.weight, hb11 / hb12 their biases (or NULL); dz is formed per group from the rows the kernel holds and never
    // stored; hpart [grid][4 Fin + 4 + Fout] receives one partial [dw11 | dw12 | db11 | db12 | dcb] per workgroup (gml_ml3_split_bwd's order)
    const float* hb11;
    const float* hb12;
    float* hpart;
#ifdef GML_BWD2_TIMING
    unsigned long long* prof;    // debug build: per-phase cycle sums
#endif
};

// static bounds of the register-batched staging (groups beyond them use the rolled loops)
#define GML_BWD_ECAP_MAX 1024
#define GML_BWD_XCAP_MAX 192

template <int S, int NFB, int NOB>
struct GmlBwdCfg {
    static constexpr int FINP = NFB * 16, FOUTP = NOB * 16;
    static constexpr int LDW = FOUTP + 1;       // W_l[s][f][LDW]: both MFMA read patterns <= 2-way conflicts
    static constexpr int LDG = FOUTP + 4;       // G window rows (b128 aligned)
    static constexpr int LDP = FOUTP + 4;       // P exchange rows (b128 aligned)
    static constexpr int NBLK = S * NFB * NOB;  // dW output blocks of 16x16
    // supports exchanged per slab: smallest SE | S with SE*NFB*NOB a multiple of 4 (one block per wave per step)
    static constexpr int se_() {
        for (int se = 1; se <= 4; ++se)
            if (S % se == 0 && (se * NFB * NOB) % 4 == 0) return se;
        return 0;
    }
    static constexpr int SE = se_();
    static constexpr bool OK = SE > 0;
    static constexpr int IPS = OK ? SE * NFB * NOB / 4 : 1;   // blocks per wave per slab
    static constexpr int NSLAB = OK ? S / SE : 1;
    static constexpr int NACC = IPS * NSLAB;                  // persistent dW accumulators (f32x4) per wave
    static constexpr int W_FLOATS = (S * FINP * LDW + 3) / 4 * 4;
    static constexpr int RP_FLOATS = 68;
    __host__ __device__ static constexpr int pex_floats() { return 2 * SE * 64 * LDP; }
    __host__ __device__ static int gs_floats(int xcap) {
        const int a = xcap * LDG, b = pex_floats();
        return a > b ? a : b;
    }
    __host__ __device__ static size_t lds_bytes(int ecap, int xcap) {
        return sizeof(float) * (size_t)(W_FLOATS + RP_FLOATS + ecap + ecap * S + gs_floats(xcap));
    }
};

template <int S, int NFB, int NOB>
__global__ __launch_bounds__(256, 2) void gml_k_spectconv_bwd(const GmlBwdParams p) {
    using C = GmlBwdCfg<S, NFB, NOB>;
    constexpr int FINP = C::FINP, LDW = C::LDW, LDG = C::LDG, LDP = C::LDP;
    constexpr int KF = FINP / 4;
    constexpr int VAL_ALIGN = (S % 4 == 0) ? 4 : ((S % 2 == 0) ? 2 : 1);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W_l = lds;
    int* rp_l = reinterpret_cast<int*>(W_l + C::W_FLOATS);
    int* col_l = rp_l + C::RP_FLOATS;
    float* ea_l = reinterpret_cast<float*>(col_l + p.ecap);
    float* gs = ea_l + (size_t)p.ecap * S;
    float* pex = gs;                                        // aliases the G window once the edge phase is over

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);

    // W[s][f][o] -> W_l[(s*FINP + f)*LDW + o], zero padded
    for (int e = tid; e < S * FINP * C::FOUTP; e += 256) {
        const int o = e % C::FOUTP, f = (e / C::FOUTP) % FINP, s = e / (C::FOUTP * FINP);
        float v = 0.f;
        if (f < p.Fin && o < p.Fout) v = p.w[((int64_t)s * p.Fin + f) * p.Fout + o];
        W_l[(s * FINP + f) * LDW + o] = v;
    }

    f32x4 dwacc[C::NACC];
#pragma unroll
    for (int i = 0; i < C::NACC; ++i) dwacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int g = g0; g < g1; ++g) {
        const int64_t r0 = (int64_t)g * 64;
        const int nr = (int)min((int64_t)64, p.nrows - r0);
        const int4 gi = *reinterpret_cast<const int4*>(p.ginfo + (int64_t)g * GML_GREC_INTS(64));   // row order unused here
        const int kb = gi.x, ne = gi.y, lo = gi.z, nwin = gi.w;
        __syncthreads();                                     // previous group is done with every LDS region

        // ---- stage (coalesced): rowptr slice, local column ids, value rows, G window.
        // Fast path: every global load of the group is issued before the first LDS write (registers are free at
        // this point: Z / P are dead), so the group pays ONE memory latency.  A rolled `lds[i] = g[i]` loop
        // compiles to load / s_waitcnt vmcnt(0) / ds_write per iteration, i.e. a latency per element.
        if (tid <= nr) rp_l[tid] = p.rowptr[r0 + tid];
        if ((S % 4 == 0) && p.gvec && ne <= GML_BWD_ECAP_MAX && nwin <= GML_BWD_XCAP_MAX) {
            constexpr int NC = GML_BWD_ECAP_MAX / 256, NE4 = GML_BWD_ECAP_MAX * (S / 4) / 256;
            constexpr int NG4 = (GML_BWD_XCAP_MAX * (C::FOUTP / 4) + 255) / 256;
            int cv[NC];
            f32x4 ev4[NE4], gv4[NG4];
            const f32x4* src = reinterpret_cast<const f32x4*>(p.val + (int64_t)kb * S);
#pragma unroll
            for (int t = 0; t < NC; ++t) { const int i = tid + 256 * t; cv[t] = (i < ne) ? p.col[kb + i] : 0; }
#pragma unroll
            for (int t = 0; t < NE4; ++t) {
                const int i = tid + 256 * t;
                ev4[t] = (i < ne * (S / 4)) ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < NG4; ++t) {
                const int i = tid + 256 * t;
                const int rr = i / (C::FOUTP / 4), o4 = (i % (C::FOUTP / 4)) * 4;
                gv4[t] = (i < nwin * (C::FOUTP / 4) && o4 < p.Fout)
                             ? *reinterpret_cast<const f32x4*>(p.g + (int64_t)(lo + rr) * p.ldg + o4)
                             : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < NC; ++t) { const int i = tid + 256 * t; if (i < ne) col_l[i] = cv[t] - lo; }
#pragma unroll
            for (int t = 0; t < NE4; ++t) {
                const int i = tid + 256 * t;
                if (i < ne * (S / 4)) reinterpret_cast<f32x4*>(ea_l)[i] = ev4[t];
            }
#pragma unroll
            for (int t = 0; t < NG4; ++t) {
                const int i = tid + 256 * t;
                const int rr = i / (C::FOUTP / 4), o4 = (i % (C::FOUTP / 4)) * 4;
                if (i < nwin * (C::FOUTP / 4)) *reinterpret_cast<f32x4*>(gs + rr * LDG + o4) = gv4[t];
            }
        } else {
            for (int i = tid; i < ne; i += 256) col_l[i] = p.col[kb + i] - lo;
            if constexpr (S % 4 == 0) {
                const f32x4* src = reinterpret_cast<const f32x4*>(p.val + (int64_t)kb * S);
                for (int i = tid; i < ne * (S / 4); i += 256) reinterpret_cast<f32x4*>(ea_l)[i] = src[i];
            } else {
                for (int i = tid; i < ne * S; i += 256) ea_l[i] = p.val[(int64_t)kb * S + i];
            }
            if (p.gvec) {                                    // rows padded to a float4 multiple, 16-B aligned
                for (int i = tid; i < nwin * (C::FOUTP / 4); i += 256) {
                    const int rr = i / (C::FOUTP / 4), o4 = (i % (C::FOUTP / 4)) * 4;
                    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (o4 < p.Fout) t = *reinterpret_cast<const f32x4*>(p.g + (int64_t)(lo + rr) * p.ldg + o4);   // pad cols are 0
                    *reinterpret_cast<f32x4*>(gs + rr * LDG + o4) = t;
                }
            } else {
                for (int i = tid; i < nwin * C::FOUTP; i += 256) {
                    const int rr = i / C::FOUTP, o = i % C::FOUTP;
                    gs[rr * LDG + o] = (o < p.Fout) ? p.g[(int64_t)(lo + rr) * p.ldg + o] : 0.f;
                }
            }
        }
        __syncthreads();

        const int row = wave * 16 + r16;                     // row of the group owned by this lane
        const bool rvalid = row < nr;
        const int kbeg = rvalid ? rp_l[row] - kb : 0;
        const int kend = rvalid ? rp_l[row + 1] - kb : 0;

        // ---- Z^T = W^T X^T : lane (r16, kq) gets Z[row][s][ob*16 + 4*kq + reg]
        f32x2 Z[S][NOB][2], P[S][NOB][2];        // o = ob*16 + 4*kq + 2*h + {x, y}: aligned register pairs for v_pk_fma_f32
        {
            // own X row, KF consecutive features per lane: the contraction index of step t is f = KF*kq + t
            float xb[KF];
            {
                const float* xr = p.x + (r0 + row) * p.ldx + KF * kq;
                if (p.xvec && KF % 4 == 0) {
#pragma unroll
                    for (int q4 = 0; q4 < KF / 4; ++q4) {
                        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (rvalid && KF * kq + 4 * q4 < p.Fin) t = *reinterpret_cast<const f32x4*>(xr + 4 * q4);
                        xb[4 * q4] = t.x; xb[4 * q4 + 1] = t.y; xb[4 * q4 + 2] = t.z; xb[4 * q4 + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < KF; ++t) xb[t] = (rvalid && KF * kq + t < p.Fin) ? xr[t] : 0.f;
                }
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f32x4 d[NOB];                                // NOB independent accumulator chains, interleaved
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) d[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < KF; ++t)
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) {
                        const float a = W_l[(s * FINP + KF * kq + t) * LDW + ob * 16 + r16];
                        d[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[t], d[ob], 0, 0, 0);
                    }
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        Z[s][ob][h] = f32x2{d[ob][2 * h], d[ob][2 * h + 1]};
                        P[s][ob][h] = f32x2{0.f, 0.f};
                    }
            }
        }

        // ---- edge phase
        for (int k = kbeg; k < kend; ++k) {
            const int dstl = col_l[k];
            float ev[S];
            gml_load_row<S, VAL_ALIGN>(ea_l + k * S, ev);
            f32x2 gv[NOB][2];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(gs + dstl * LDG + ob * 16 + 4 * kq);
                gv[ob][0] = f32x2{t.x, t.y};
                gv[ob][1] = f32x2{t.z, t.w};
            }
            // packed math (v_pk_fma_f32): both the P update and the Z.g dot run on float2 lanes; the dot keeps
            // two partial sums (even / odd o) that are added once at the end
            float d[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f32x2 a2 = f32x2{0.f, 0.f};
                const f32x2 e2 = f32x2{ev[s], ev[s]};
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        P[s][ob][h] = e2 * gv[ob][h] + P[s][ob][h];
                        a2 = Z[s][ob][h] * gv[ob][h] + a2;
                    }
                d[s] = a2.x + a2.y;
            }
            // the 4 lanes of a row (kq = 0..3) each hold a quarter of the o-sum: v_permlane16_swap /
            // v_permlane32_swap fold four values at a time so that lane kq ends up with the total of
            // value 4c + kq (6 VALU ops per 4 values, no LDS round trip); value row k is consumed, so
            // dval overwrites it in place.
#pragma unroll
            for (int c = 0; c < (S + 3) / 4; ++c) {
                const float v0 = d[4 * c], v1 = (4 * c + 1 < S) ? d[4 * c + 1] : 0.f;
                const float v2 = (4 * c + 2 < S) ? d[4 * c + 2] : 0.f, v3 = (4 * c + 3 < S) ? d[4 * c + 3] : 0.f;
                const auto a01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
                const auto a23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
                const float c01 = __uint_as_float(a01[0]) + __uint_as_float(a01[1]);
                const float c23 = __uint_as_float(a23[0]) + __uint_as_float(a23[1]);
                const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(c01), __float_as_uint(c23), false, false);
                const float tot = __uint_as_float(b[0]) + __uint_as_float(b[1]);
                if (4 * c + kq < S) ea_l[k * S + 4 * c + kq] = tot;
            }
        }
        __syncthreads();                                     // dval rows complete; G window no longer needed

        if (p.dval) {
            if constexpr (S % 4 == 0) {
                f32x4* dst = reinterpret_cast<f32x4*>(p.dval + (int64_t)kb * S);
                for (int i = tid; i < ne * (S / 4); i += 256) dst[i] = reinterpret_cast<const f32x4*>(ea_l)[i];
            } else {
                for (int i = tid; i < ne * S; i += 256) p.dval[(int64_t)kb * S + i] = ea_l[i];
            }
        }

        // ---- dX = P W^T : contraction over (s, o), P registers are the A fragments
        if (p.dx) {
            f32x4 dxa[NFB];                                  // NFB independent accumulator chains, interleaved
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) dxa[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int fb = 0; fb < NFB; ++fb) {
                            const float b = W_l[(s * FINP + fb * 16 + r16) * LDW + ob * 16 + 4 * kq + i];
                            dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(P[s][ob][i >> 1][i & 1], b, dxa[fb], 0, 0, 0);
                        }
            if (p.flags & GML_ACCUM) {                       // own uniform branch: see the note in the forward epilogue
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    const int f = fb * 16 + r16;
                    float old[4];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int lr = wave * 16 + 4 * kq + reg;
                        old[reg] = (f < p.Fin && lr < nr) ? p.dx[(r0 + lr) * p.lddx + f] : 0.f;
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int lr = wave * 16 + 4 * kq + reg;
                        if (f < p.Fin && lr < nr) p.dx[(r0 + lr) * p.lddx + f] = dxa[fb][reg] + old[reg];
                    }
                }
            } else {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    const int f = fb * 16 + r16;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int lr = wave * 16 + 4 * kq + reg;
                        if (f < p.Fin && lr < nr) p.dx[(r0 + lr) * p.lddx + f] = dxa[fb][reg];
                    }
                }
            }
        }

        // ---- dW += X^T P : contraction over the 64 rows of the group, slab of SE supports at a time
        if (p.dw_partial) {
            // A fragments (X^T of the group's 64 rows) of the blocks this wave owns: loaded ONCE per group, before
            // the slab loop -- a load placed after a barrier could not be hoisted and would stall every slab
            float xa[C::IPS][16];
#pragma unroll
            for (int it = 0; it < C::IPS; ++it) {
                const int fb = ((it * 4 + wave) / NOB) % NFB;
                const int af = fb * 16 + r16;                                              // A[i = f][k = row]
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int ar = 4 * t + kq;
                    xa[it][t] = (ar < nr && af < p.Fin) ? p.x[(r0 + ar) * p.ldx + af] : 0.f;
                }
            }
#pragma unroll
            for (int sl = 0; sl < C::NSLAB; ++sl) {
                float* pb = pex + (sl & 1) * (C::SE * 64 * LDP);
#pragma unroll
                for (int se = 0; se < C::SE; ++se)
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob)
                        *reinterpret_cast<f32x4*>(pb + (se * 64 + row) * LDP + ob * 16 + 4 * kq) =
                            f32x4{P[sl * C::SE + se][ob][0].x, P[sl * C::SE + se][ob][0].y, P[sl * C::SE + se][ob][1].x,
                                  P[sl * C::SE + se][ob][1].y};
                __syncthreads();
#pragma unroll
                for (int it = 0; it < C::IPS; ++it) {
                    const int blk = it * 4 + wave;           // (se, fb, ob) block inside the slab
                    const int ob = blk % NOB, se = blk / (NOB * NFB);
                    f32x4 d0 = dwacc[sl * C::IPS + it], d1 = f32x4{0.f, 0.f, 0.f, 0.f};    // two chains: even / odd rows
#pragma unroll
                    for (int t = 0; t < 16; t += 2) {
                        const float b0 = pb[(se * 64 + 4 * t + kq) * LDP + ob * 16 + r16];       // B[k = row][j = o]
                        const float b1 = pb[(se * 64 + 4 * (t + 1) + kq) * LDP + ob * 16 + r16];
                        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[it][t], b0, d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[it][t + 1], b1, d1, 0, 0, 0);
                    }
                    dwacc[sl * C::IPS + it] = d0 + d1;
                }
                // a slab buffer is rewritten two slabs later: every wave has passed the barrier in between
            }
        }
    }

    // ---- one dW partial per workgroup: block (s, fb, ob): D[i = f][j = o], lane rows 4*kq + reg, col r16
    if (p.dw_partial && g0 < g1) {
        float* out = p.dw_partial + (int64_t)wg * S * p.Fin * p.Fout;
#pragma unroll
        for (int sl = 0; sl < C::NSLAB; ++sl)
#pragma unroll
            for (int it = 0; it < C::IPS; ++it) {
                const int blk = it * 4 + wave;
                const int ob = blk % NOB, fb = (blk / NOB) % NFB, s = sl * C::SE + blk / (NOB * NFB);
                const int o = ob * 16 + r16;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int f = fb * 16 + 4 * kq + reg;
                    if (f < p.Fin && o < p.Fout) out[((int64_t)s * p.Fin + f) * p.Fout + o] = dwacc[sl * C::IPS + it][reg];
                }
            }
    }
}

// out[j] = sum_w partial[w][j]   (fixed order -> deterministic)
__global__ void gml_k_reduce_rows(const float* __restrict__ partial, int64_t nparts, int64_t n, float* __restrict__ out);

template <int S, int NFB, int NOB>
int gml_launch_bwd(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st);

#define GML_DEFINE_BWD(SV, NFBV, NOBV)                                                                       \
    template <>                                                                                              \
    int gml_launch_bwd<SV, NFBV, NOBV>(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st) {       \
        static_assert(GmlBwdCfg<SV, NFBV, NOBV>::OK, "no slab size for this shape");                         \
        GML_ALLOW_BIG_LDS(attr_rc, (&gml_k_spectconv_bwd<SV, NFBV, NOBV>), 160 * 1024) \
        if (attr_rc != hipSuccess) return (int)attr_rc;                                                      \
        hipLaunchKernelGGL((gml_k_spectconv_bwd<SV, NFBV, NOBV>), grid, dim3(256), lds, st, p);              \
        return gml_launch_status();                                                                          \
    }
