// Fused multi-support spectral convolution for gfx950:
//
//   out[r, :] (op)= act( sum_s ( sum_{k in row r} val[pos(k), s] * x[col[k], :] ) @ W[s] + bias )
//
// replaces, in one launch, the S x (index_select, mul, scatter_add, matmul, add) of
// /root/reference/libs/spect_conv.py:76-80 + bias :93-94 (+ the relu of ML3Layer :209).
//
// Work decomposition (wave64, one 16-row tile per wave, 4 waves per workgroup):
//   lane l = (r16 = l & 15, kq = l >> 4) owns output-row r16 of the tile and, inside a chunk of
//   CH = 4*FPL input features, the FPL consecutive features [kq*FPL, kq*FPL+FPL).
//   * aggregation (VALU): the lane walks the CSR row in order (= the reference's per-target
//     summation order) and keeps acc[s][j] = H[r16][s][f] in registers - H never goes to HBM;
//   * projection (MFMA): acc[s][j] IS the A fragment of v_mfma_f32_16x16x4_f32
//     (A[i = l&15][k = l>>4]); the K order of the contraction is permuted to
//     (chunk, s, j, kq) and W is laid out in LDS once per workgroup in exactly that order, one
//     dword per lane per MFMA, conflict-free: Wl[(s*FPL + j)*NB + nb][lane];
//   * epilogue: D[row = 4*(l>>4) + reg][col = l&15] (+ old out) + bias, relu, 64-B row segments.
// f32-in MFMA is bit-identical to an fmaf chain (exact fp32), so parity is fp32-roundoff class.
#pragma once
#include "gml_common.h"

struct GmlFwdParams {
    const int32_t* rowptr;
    const int32_t* col;
    const int32_t* epos;
    const float* val;
    const float* x;
    int64_t ldx;
    const float* w;
    int64_t w_ss, w_si, w_so;
    const float* bias;
    float* out;
    int64_t ldo;
    int64_t nrows;
    int32_t S, Fin, Fout;
    uint32_t flags;
    int32_t s0;          // first support handled by this launch
    int32_t npass;       // supports handled = npass * SC
    int32_t nchunks;     // ceil(Fin / (4*FPL))
    int32_t ngroups;     // ceil(nrows / 64): one workgroup step = 4 tiles of 16 rows
    int32_t groups_per_wg;
    int32_t allw;        // whole W of this launch resident in LDS
    int32_t val_vec;     // value rows (S floats apart, starting at s0) keep the SC alignment class
    int32_t wfloats;     // floats of LDS reserved for W in front of the staging area
    const int32_t* ginfo; // [ngroups][GML_GREC_INTS(64)] records of gml_csr_group_info (64-row groups)
    // Hadamard branch fused into the 8-wave kernel (gml_ml3_fwd): out[r, mix_col + o] = tanh(x w11_o + b11_o) tanh(x w12_o + b12_o)
    const float* w11; const float* b11; const float* w12; const float* b12;
    int32_t F2, mix_col;
    float* hout;          // stand-alone SpMM on the 8-wave kernel: H [N, S, Fin] receives the aggregate, no projection
    unsigned long long* prof;   // timing build (-DGML_FWD2_TIMING): per-phase cycle sums
    int32_t nw;           // 8-wave kernel family: waves per workgroup (8: 128-row groups, 4: ranked 64-row groups)
    // epilogues of the ring kernel (gml_spectconv_fwd_epi): 0 = sum_s H_s W_s; 1 = SpectConCatConv: H_s W_s written to column block
    // s + cc_off (libs/spect_conv.py:137-158); 2 = depthwise: (sum_s ds[s] . H_s + ds[S] . x) W_0 (libs/spect_conv.py:81-91)
    const float* ds;      // depthwise: [S (+ 1 if ds_self), Fin] scales (row 0 already holds 1 + DSweight[0])
    int32_t epl, ds_self, cc_off;
};

// Per-group staging capacities.  A group = 64 consecutive output rows = the 4 tiles a workgroup
// works on at once.  Its CSR slice (column ids + value rows) and the window of X rows its columns
// fall into are copied to LDS with coalesced loads, so the per-edge gathers of the aggregation are
// LDS reads (one HBM/L2 latency per group instead of two dependent ones per edge).  Batches of small
// graphs are block diagonal, so the window is ~64 + 2*max_graph_size rows.  A group that does not
// fit (more edges than ECAP, wider window than XCAP) takes the global-gather path -- same results.
#define GML_GROUP 64
#define GML_ECAP 512
#define GML_XCAP 144

template <int SC, int FPL>
struct GmlStage {
    static constexpr int CH = 4 * FPL;
    // staged edges per 64-row group: 8 per row; 16 for the 6-support chunk -- sr25.py's supports (S = 6) have 13 entries per row
    // and every one of its groups used to fall back to global gathers
    static constexpr int ECAP = SC == 6 ? 2 * GML_ECAP : GML_ECAP;
    static constexpr int LDX = CH + 4;                       // +16 B: spreads rows over LDS banks, keeps b128 alignment
    static constexpr int RP = 76;                            // rowptr slice [0..64], min scratch [66..69], max scratch [70..73]
    static constexpr int OFF_COL = RP;
    static constexpr int OFF_EA = OFF_COL + ECAP;
    static constexpr int OFF_X = OFF_EA + ECAP * SC;
    static constexpr int FLOATS = OFF_X + GML_XCAP * LDX;
};

template <int SC, int FPL, int NB>
__device__ __forceinline__ void gml_stage_w(float* __restrict__ dst, const GmlFwdParams& p, int pass, int c) {
    constexpr int CH = 4 * FPL;
    // walk W in memory order of its two inner dims (coalesced when w_so == 1 or w_si == 1)
    const bool o_fast = p.w_so <= p.w_si;
    const int nf = min(CH, p.Fin - c * CH);
    const int no = p.Fout;
    for (int s = 0; s < SC; ++s) {
        const float* ws = p.w + (int64_t)(p.s0 + pass * SC + s) * p.w_ss;
        for (int e = threadIdx.x; e < CH * NB * 16; e += blockDim.x) {
            int fl, o;
            if (o_fast) { fl = e / (NB * 16); o = e % (NB * 16); }
            else { o = e / CH; fl = e % CH; }
            float v = 0.f;
            if (fl < nf && o < no) v = ws[(int64_t)(c * CH + fl) * p.w_si + (int64_t)o * p.w_so];
            const int kq = fl / FPL, j = fl % FPL;
            dst[((s * FPL + j) * NB + (o >> 4)) * 64 + kq * 16 + (o & 15)] = v;
        }
    }
}

// bf16x3 variant (FPL == 8): the W block is stored as MFMA B fragments of v_mfma_f32_16x16x32_bf16 -- 8 bf16
// (16 bytes) per lane, k = 8*(lane>>4) + i <-> feature c*32 + 8*kq + i, column nb*16 + (lane&15) -- once as the
// high parts and once as the low parts.  Same LDS footprint as the fp32 block (SC*NB*2 KiB).
template <int SC, int NB>
__device__ __forceinline__ void gml_stage_w_bf16(bf16x8* __restrict__ dst, const GmlFwdParams& p, int pass, int c) {
    for (int e = threadIdx.x; e < SC * NB * 64; e += blockDim.x) {
        const int lane = e & 63, nb = (e >> 6) % NB, s = (e >> 6) / NB;
        const int o = nb * 16 + (lane & 15);
        const float* ws = p.w + (int64_t)(p.s0 + pass * SC + s) * p.w_ss + (int64_t)o * p.w_so;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = c * 32 + 8 * (lane >> 4) + i;
            v[i] = (f < p.Fin && o < p.Fout) ? ws[(int64_t)f * p.w_si] : 0.f;
        }
        bf16x8 hi, lo;
        gml_split8(v, hi, lo);
        dst[e] = hi;
        dst[SC * NB * 64 + e] = lo;
    }
}

// Per-thread registers that carry the NEXT group's CSR slice / value rows / X window while the
// current group is being computed (global -> reg early, reg -> LDS late: the load latency hides
// behind the aggregation + MFMA of the current group instead of stalling every group).
template <int SC, int FPL>
struct GmlPrefetch {
    using ST = GmlStage<SC, FPL>;
    static constexpr int CH = ST::CH;
    static constexpr int CL = ST::ECAP / 256;
    static constexpr int EN = ST::ECAP * SC / 256;                               // floats of value rows per thread
    static constexpr int XN4 = (GML_XCAP * (CH / 4) + 255) / 256;                // float4 of X window per thread
    static constexpr int XN = (GML_XCAP * CH + 255) / 256 > 4 * XN4 ? (GML_XCAP * CH + 255) / 256 : 4 * XN4;
    int kb, ne, lo, nwin, nr;
    bool staged;
    int rp;
    int colv[CL];
    float ev[EN];
    float xv[XN];
};

template <int SC, int FPL, bool XVEC>
__device__ __forceinline__ void gml_prefetch_issue(GmlPrefetch<SC, FPL>& q, const GmlFwdParams& p, int g, int tid) {
    using PF = GmlPrefetch<SC, FPL>;
    constexpr int CH = PF::CH;
    const int64_t r0 = (int64_t)g * GML_GROUP;
    q.nr = (int)min((int64_t)GML_GROUP, p.nrows - r0);
    const int32_t* rec = p.ginfo + (int64_t)g * GML_GREC_INTS(GML_GROUP);
    const int4 gi = *reinterpret_cast<const int4*>(rec);                         // {kb, ne, lo, nwin}
    q.kb = gi.x; q.ne = gi.y; q.lo = gi.z; q.nwin = gi.w;
    q.staged = (q.ne <= PF::ST::ECAP) && (q.nwin <= GML_XCAP) && (p.epos == nullptr);
    q.rp = (tid <= q.nr) ? p.rowptr[r0 + tid] : 0;
    if (!q.staged) return;
#pragma unroll
    for (int t = 0; t < PF::CL; ++t) {
        const int i = tid + 256 * t;
        q.colv[t] = (i < q.ne) ? p.col[q.kb + i] : 0;
    }
    const int sbase = p.s0;
    // NOTE: indirect value rows (epos != NULL) never take the staged path (q.staged is false for them): folding
    // `pk = epos ? epos[k] : k` into this loop makes hipcc emit a predicated load + s_waitcnt vmcnt(0) per
    // element, which drains every load already in flight and serialises the whole prefetch (+3 us per group).
    {
        const float* vb = p.val + (int64_t)q.kb * p.S + sbase;
        if (p.val_vec && (SC % 4 == 0)) {
#pragma unroll
            for (int t = 0; t < PF::EN / 4; ++t) {
                const int idx4 = tid + 256 * t;                                  // float4 index inside [ne][SC]
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx4 < q.ne * (SC / 4)) {
                    const int i = idx4 / (SC / 4), s4 = (idx4 % (SC / 4)) * 4;
                    v = *reinterpret_cast<const f32x4*>(vb + (int64_t)i * p.S + s4);
                }
                q.ev[4 * t] = v.x; q.ev[4 * t + 1] = v.y; q.ev[4 * t + 2] = v.z; q.ev[4 * t + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < PF::EN; ++t) {
                const int idx = tid + 256 * t;
                float v = 0.f;
                if (idx < q.ne * SC) v = vb[(int64_t)(idx / SC) * p.S + idx % SC];
                q.ev[t] = v;
            }
        }
    }
    if constexpr (XVEC) {
#pragma unroll
        for (int t = 0; t < PF::XN4; ++t) {
            const int idx4 = tid + 256 * t;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx4 < q.nwin * (CH / 4)) {
                const int rr = idx4 / (CH / 4), f4 = (idx4 % (CH / 4)) * 4;
                if (f4 < p.Fin) v = *reinterpret_cast<const f32x4*>(p.x + (int64_t)(q.lo + rr) * p.ldx + f4);
            }
            q.xv[4 * t] = v.x; q.xv[4 * t + 1] = v.y; q.xv[4 * t + 2] = v.z; q.xv[4 * t + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int t = 0; t < (GML_XCAP * CH + 255) / 256; ++t) {
            const int idx = tid + 256 * t;
            float v = 0.f;
            if (idx < q.nwin * CH) {
                const int rr = idx / CH, f = idx % CH;
                if (f < p.Fin) v = p.x[(int64_t)(q.lo + rr) * p.ldx + f];
            }
            q.xv[t] = v;
        }
    }
}

template <int SC, int FPL, bool XVEC>
__device__ __forceinline__ void gml_prefetch_commit(const GmlPrefetch<SC, FPL>& q, const GmlFwdParams& p, int tid,
                                                    int* rp_l, int* col_l, float* ea_l, float* xs) {
    using PF = GmlPrefetch<SC, FPL>;
    constexpr int CH = PF::CH;
    constexpr int LDX = PF::ST::LDX;
    if (tid <= q.nr) rp_l[tid] = q.rp;
    if (!q.staged) return;
#pragma unroll
    for (int t = 0; t < PF::CL; ++t) {
        const int i = tid + 256 * t;
        if (i < q.ne) col_l[i] = q.colv[t] - q.lo;
    }
    if (p.val_vec && (SC % 4 == 0)) {
#pragma unroll
        for (int t = 0; t < PF::EN / 4; ++t) {
            const int idx4 = tid + 256 * t;
            if (idx4 < q.ne * (SC / 4))
                *reinterpret_cast<f32x4*>(ea_l + 4 * idx4) = f32x4{q.ev[4 * t], q.ev[4 * t + 1], q.ev[4 * t + 2], q.ev[4 * t + 3]};
        }
    } else {
#pragma unroll
        for (int t = 0; t < PF::EN; ++t) {
            const int idx = tid + 256 * t;
            if (idx < q.ne * SC) ea_l[idx] = q.ev[t];
        }
    }
    if constexpr (XVEC) {
#pragma unroll
        for (int t = 0; t < PF::XN4; ++t) {
            const int idx4 = tid + 256 * t;
            if (idx4 < q.nwin * (CH / 4)) {
                const int rr = idx4 / (CH / 4), f4 = (idx4 % (CH / 4)) * 4;
                *reinterpret_cast<f32x4*>(xs + rr * LDX + f4) = f32x4{q.xv[4 * t], q.xv[4 * t + 1], q.xv[4 * t + 2], q.xv[4 * t + 3]};
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < (GML_XCAP * CH + 255) / 256; ++t) {
            const int idx = tid + 256 * t;
            if (idx < q.nwin * CH) xs[(idx / CH) * LDX + (idx % CH)] = q.xv[t];
        }
    }
}

template <int SC, int FPL, int NB, bool XVEC, bool BF>
__global__ __launch_bounds__(256, 2) void gml_k_spectconv_fwd(const GmlFwdParams p) {
    static_assert(!BF || FPL == 8, "the bf16x3 projection needs 8 features per lane (K = 32 per MFMA)");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using ST = GmlStage<SC, FPL>;
    constexpr int CH = 4 * FPL;
    constexpr int LDX = ST::LDX;
    constexpr int WBLK = SC * FPL * NB * 64;
    constexpr int VAL_ALIGN = (SC % 4 == 0) ? 4 : ((SC % 2 == 0) ? 2 : 1);

    float* lds_w = lds;
    float* stg = lds + p.wfloats;
    int* rp_l = reinterpret_cast<int*>(stg);                 // [0..64] rowptr slice of the group
    int* col_l = reinterpret_cast<int*>(stg + ST::OFF_COL);  // window-local column ids
    float* ea_l = stg + ST::OFF_EA;                          // [ne][SC] value rows of the current pass
    float* xs = stg + ST::OFF_X;                             // [nwin][LDX] X window, current feature chunk

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);
    if (g0 >= g1) return;

    GmlPrefetch<SC, FPL> q;
    gml_prefetch_issue<SC, FPL, XVEC>(q, p, g0, tid);

    if (p.allw) {
        for (int pass = 0; pass < p.npass; ++pass)
            for (int c = 0; c < p.nchunks; ++c) {
                if constexpr (BF) gml_stage_w_bf16<SC, NB>(reinterpret_cast<bf16x8*>(lds_w + (pass * p.nchunks + c) * WBLK), p, pass, c);
                else gml_stage_w<SC, FPL, NB>(lds_w + (pass * p.nchunks + c) * WBLK, p, pass, c);
            }
    }
    float bias_r[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias_r[nb] = (p.bias && nb * 16 + r16 < p.Fout) ? p.bias[nb * 16 + r16] : 0.f;

    for (int g = g0; g < g1; ++g) {
        const int64_t r0 = (int64_t)g * GML_GROUP;
        const int nr = q.nr, kb = q.kb, ne = q.ne, lo = q.lo, nwin = q.nwin;
        const bool staged = q.staged;
        gml_prefetch_commit<SC, FPL, XVEC>(q, p, tid, rp_l, col_l, ea_l, xs);
        __syncthreads();
        if (g + 1 < g1) gml_prefetch_issue<SC, FPL, XVEC>(q, p, g + 1, tid);   // in flight during this group's compute

        const int tile_row = wave * 16 + r16;                // row inside the group owned by this lane
        const bool rvalid = tile_row < nr;
        const int kbeg = rvalid ? rp_l[tile_row] : 0;
        const int kend = rvalid ? rp_l[tile_row + 1] : 0;

        f32x4 oacc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int pass = 0; pass < p.npass; ++pass) {
            const int sbase = p.s0 + pass * SC;
            for (int c = 0; c < p.nchunks; ++c) {
                const bool first = (pass + c) == 0;
                const float* wl = lds_w;
                if (p.allw) wl = lds_w + (pass * p.nchunks + c) * WBLK;
                if (!first || !p.allw) {
                    __syncthreads();                         // LDS regions about to be overwritten are idle
                    if (!p.allw) {
                        if constexpr (BF) gml_stage_w_bf16<SC, NB>(reinterpret_cast<bf16x8*>(lds_w), p, pass, c);
                        else gml_stage_w<SC, FPL, NB>(lds_w, p, pass, c);
                    }
                    if (staged && !first) {
                        if (c == 0) {                        // value rows of this pass: [ne][SC]
                            for (int idx = tid; idx < ne * SC; idx += 256)     // staged => epos == NULL
                                ea_l[idx] = p.val[(int64_t)(kb + idx / SC) * p.S + sbase + idx % SC];
                        }
                        for (int idx = tid; idx < nwin * CH; idx += 256) {
                            const int rr = idx / CH, f = idx % CH;
                            const int gf = c * CH + f;
                            xs[rr * LDX + f] = (gf < p.Fin) ? p.x[(int64_t)(lo + rr) * p.ldx + gf] : 0.f;
                        }
                    }
                    __syncthreads();
                }

                float acc[SC][FPL];
#pragma unroll
                for (int s = 0; s < SC; ++s)
#pragma unroll
                    for (int j = 0; j < FPL; ++j) acc[s][j] = 0.f;

#ifdef GML_ABLATE
                if (p.flags & 0x200u) {
                    // ablation: no aggregation
                } else
#endif
                if (staged) {
                    for (int k = kbeg - kb; k < kend - kb; ++k) {
                        const int srcl = col_l[k];
                        float ev[SC], xv[FPL];
                        gml_load_row<SC, VAL_ALIGN>(ea_l + k * SC, ev);
                        gml_load_row<FPL, 4>(xs + srcl * LDX + kq * FPL, xv);
#pragma unroll
                        for (int s = 0; s < SC; ++s)
#pragma unroll
                            for (int j = 0; j < FPL; ++j) acc[s][j] = fmaf(ev[s], xv[j], acc[s][j]);
                    }
                } else {
                    const int f0 = c * CH + kq * FPL;
                    for (int k = kbeg; k < kend; ++k) {
                        const int src = p.col[k];
                        const int64_t pk = p.epos ? (int64_t)p.epos[k] : (int64_t)k;
                        float ev[SC];
                        if (p.val_vec) gml_load_row<SC, VAL_ALIGN>(p.val + pk * p.S + sbase, ev);
                        else gml_load_row<SC, 1>(p.val + pk * p.S + sbase, ev);
                        float xv[FPL];
                        const float* xr = p.x + (int64_t)src * p.ldx + f0;
                        if constexpr (XVEC) {
#pragma unroll
                            for (int q4 = 0; q4 < FPL / 4; ++q4) {
                                f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                                if (f0 + 4 * q4 < p.Fin) t = *reinterpret_cast<const f32x4*>(xr + 4 * q4);
                                xv[4 * q4] = t.x; xv[4 * q4 + 1] = t.y; xv[4 * q4 + 2] = t.z; xv[4 * q4 + 3] = t.w;
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < FPL; ++j) xv[j] = (f0 + j < p.Fin) ? xr[j] : 0.f;
                        }
#pragma unroll
                        for (int s = 0; s < SC; ++s)
#pragma unroll
                            for (int j = 0; j < FPL; ++j) acc[s][j] = fmaf(ev[s], xv[j], acc[s][j]);
                    }
                }

#ifdef GML_ABLATE
                if (p.flags & 0x100u) {
                    // ablation: no projection (keep the aggregation alive).  NOTE: never ship inline asm next to
                    // the MFMA chain -- it hides the MFMA->accvgpr_read hazard from hipcc's nop insertion.
#pragma unroll
                    for (int s = 0; s < SC; ++s)
#pragma unroll
                        for (int j = 0; j < FPL; ++j) asm volatile("" ::"v"(acc[s][j]));
                } else
#endif
                {
                    if constexpr (BF) {
                        // acc[s][0..7] are exactly the 8 k-values of this lane's A fragment (k = 8*kq + j)
                        const bf16x8* wh = reinterpret_cast<const bf16x8*>(wl);
                        const bf16x8* wlo = wh + SC * NB * 64;
#pragma unroll
                        for (int s = 0; s < SC; ++s) {
                            bf16x8 ah, al;
                            gml_split8(acc[s], ah, al);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                const bf16x8 bh = wh[(s * NB + nb) * 64 + lane];
                                const bf16x8 bl = wlo[(s * NB + nb) * 64 + lane];
                                oacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, oacc[nb], 0, 0, 0);
                                oacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, oacc[nb], 0, 0, 0);
                                oacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, oacc[nb], 0, 0, 0);
                            }
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < SC; ++s)
#pragma unroll
                            for (int j = 0; j < FPL; ++j) {
                                const float a = acc[s][j];
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb) {
                                    const float b = wl[((s * FPL + j) * NB + nb) * 64 + lane];
                                    oacc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, oacc[nb], 0, 0, 0);
                                }
                            }
                    }
                }
            }
        }

#ifdef GML_ABLATE
        if (!(p.flags & 0x800u))
#endif
        {
            // The accumulate variant lives in its own uniform branch: a per-element `if (accum) v += *dst` becomes a
            // predicated load + s_waitcnt vmcnt(0) in front of EVERY store, and vmcnt also counts stores, so the
            // stores of a tile would retire one round trip at a time.
            const bool relu = (p.flags & GML_RELU) != 0;
            if (p.flags & GML_ACCUM) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int o = nb * 16 + r16;
                    float old[4];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int lr = wave * 16 + 4 * kq + reg;
                        old[reg] = (o < p.Fout && lr < nr) ? p.out[(r0 + lr) * p.ldo + o] : 0.f;
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int lr = wave * 16 + 4 * kq + reg;
                        float v = oacc[nb][reg] + bias_r[nb] + old[reg];
                        if (relu) v = fmaxf(v, 0.f);
                        if (o < p.Fout && lr < nr) p.out[(r0 + lr) * p.ldo + o] = v;
                    }
                }
            } else {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int o = nb * 16 + r16;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int lr = wave * 16 + 4 * kq + reg;
                        float v = oacc[nb][reg] + bias_r[nb];
                        if (relu) v = fmaxf(v, 0.f);
                        if (o < p.Fout && lr < nr) p.out[(r0 + lr) * p.ldo + o] = v;
                    }
                }
            }
        }
        __syncthreads();                                     // this group's LDS reads are done
    }
}

// one (SC, FPL) family: NB in {1,2,4,8}, XVEC in {0,1}
template <int SC, int FPL>
int gml_launch_fwd_family(const GmlFwdParams& p, int NB, bool xvec, bool bf, dim3 grid, size_t lds, hipStream_t st);

#define GML_FWD_GO(NBV, XV, BFV)                                                                      \
    {                                                                                                 \
        GML_ALLOW_BIG_LDS(attr_rc, (&gml_k_spectconv_fwd<SC, FPL, NBV, XV, BFV>), 160 * 1024) \
        if (attr_rc != hipSuccess) return (int)attr_rc;                                               \
        hipLaunchKernelGGL((gml_k_spectconv_fwd<SC, FPL, NBV, XV, BFV>), grid, dim3(256), lds, st, p); \
        return gml_launch_status();                                                                   \
    }
#define GML_FWD_CASE(NBV, XV)                                                                         \
    if constexpr (FPL == 8) {                                                                         \
        if (bf) GML_FWD_GO(NBV, XV, true)                                                             \
    }                                                                                                 \
    GML_FWD_GO(NBV, XV, false)

#define GML_DEFINE_FWD_FAMILY(SCV, FPLV)                                                              \
    template <>                                                                                       \
    int gml_launch_fwd_family<SCV, FPLV>(const GmlFwdParams& p, int NB, bool xvec, bool bf, dim3 grid, \
                                         size_t lds, hipStream_t st) {                                \
        constexpr int SC = SCV, FPL = FPLV;                                                           \
        if (xvec) {                                                                                   \
            switch (NB) {                                                                             \
                case 1: { GML_FWD_CASE(1, true) }                                                     \
                case 2: { GML_FWD_CASE(2, true) }                                                     \
                case 4: { GML_FWD_CASE(4, true) }                                                     \
                case 8: { GML_FWD_CASE(8, true) }                                                     \
            }                                                                                         \
        } else {                                                                                      \
            switch (NB) {                                                                             \
                case 1: { GML_FWD_CASE(1, false) }                                                    \
                case 2: { GML_FWD_CASE(2, false) }                                                    \
                case 4: { GML_FWD_CASE(4, false) }                                                    \
                case 8: { GML_FWD_CASE(8, false) }                                                    \
            }                                                                                         \
        }                                                                                             \
        return GML_E_UNSUPPORTED;                                                                     \
    }
