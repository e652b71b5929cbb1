// Fused forward, 128-row / 8-wave geometry (bf16x3 projection; Fin <= 32, Fout <= 32, S in {4, 8, 12}): same function as
// gml_k_spectconv_fwd (gml_spectconv_impl.h)
//
//   out[r, :] = act( sum_s (sum_{k in row r} val[k, s] x[col[k], :]) W_s + b )
//
// organised like the backward kernel (gml_spectconv_bwd2_impl.h): one 512-thread workgroup per CU, groups of 128 target
// rows, one 16-row tile per wave, the group record's degree-ranked row order (rows of a tile run near-equal edge
// loops; rank blocks a and 7-a on the two waves of a SIMD).  With 64 accumulators per lane instead of the backward's
// 128 there is room to keep the NEXT group's CSR slice, value rows and X window in flight in registers while this
// group is aggregated and projected (unconditional, clamped loads: see the note in the kernel).
#pragma once
#include "gml_common.h"
#include "gml_spectconv_impl.h"

// ablation builds (tools/build_variant.py fw<bits> -DGML_FWABL=<bits>): results WRONG, timing says what a phase costs
//   1 = no aggregation loop, 2 = no projection, 4 = no output stores, 8 = every group loads the workgroup's first group
#ifndef GML_FWABL
#define GML_FWABL 0
#endif
// timing build (-DGML_FWD2_TIMING): every wave sums its phase durations in registers and adds them to p.prof at the end
// (0 commit, 1 barrier, 2 issue, 3 own-row loads + row bounds, 4 aggregation, 5 value gather issue, 6 projection,
//  7 output stores, 8 Hadamard branch, 9 end barrier); tools/fwd2_phases.py
#ifdef GML_FWD2_TIMING
#define GML_TF(i) do { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); tacc_[i] += t_ - tprev_; tprev_ = t_; } while (0)
#else
#define GML_TF(i)
#endif
#define GML_FWD2_ROWS 128
#define GML_FWD2_ECAP 1024      // staged edges per group
#define GML_FWD2_XCAP 208       // staged window rows of X (128 rows + 2 x the largest graph of a block-diagonal batch)

template <int S>
struct GmlFwd2Cfg {
    static constexpr int LDX = 36;                             // X window rows (floats, b128 aligned)
    static constexpr int W_HALF = S * 32 * 32;                 // bf16 elements of one (hi or lo) W image [s][o][f]
    static constexpr int TILE_BYTES = 8 * 16 * 36 * 4;         // stand-alone SpMM: one [16][36] output tile per wave
    static constexpr int W_BYTES = 2 * W_HALF * 2 > TILE_BYTES ? 2 * W_HALF * 2 : TILE_BYTES;   // (shares the W area)
    // NW = 8 waves / 128-row groups (one workgroup per CU) or NW = 4 waves / 64-row groups (two independent workgroups
    // per CU: each one's per-group latency chain -- record, data, commit, barrier -- overlaps the other's arithmetic)
    static constexpr int rows(int nw) { return 16 * nw; }
    static constexpr int ecap(int nw) { return (S > 8 ? 12 : 8) * rows(nw); }   // staged edges per group (S = 12, counting.py: 7 edges per row)
    static constexpr int xcap(int nw) { return nw == 8 ? GML_FWD2_XCAP : 144; }   // 64 rows + 2 x the largest graph
    static constexpr size_t lds_bytes(int nw = 8) {
        return (size_t)W_BYTES + (rows(nw) + 8) * 4 + (size_t)ecap(nw) * 4 + (size_t)ecap(nw) * S * 4 +
               (size_t)xcap(nw) * LDX * 4;
    }
};

// XVEC: the X rows are float4-addressable (ldx % 4 == 0, aligned base); else the window is staged element-wise
// MIX: the ML3Layer Hadamard branch (F2 <= 8 outputs) of the group's own rows rides along: one more K = 32 MFMA triple
// per tile against the [w11; w12] rows instead of a second pass over x by another kernel
// NOB = 0: stand-alone SpMM instantiation (p.hout receives the aggregate H; no W image, no projection)
// EP: value rows are gathered through p.epos (row of CSR position k = val[epos[k]]): the ML3Layer edge branch then lives in
// ONE edge order (the backward's) and writes its output once -- the second, scattered copy cost the HBM-bound edge forward
// 37 % of its time (profiles/r02_g_edge_fwd_ablation.txt).  The positions are loaded with the other prefetch loads, the
// value rows they address after the aggregation loop (the positions have arrived by then: no exposed dependent latency).
// F16 (round 6, GML_F16X3: see gml_spectconv_fwd3_impl.h): projection and Hadamard branch on f16 (hi, lo) pieces under power-of-two
// scales -- the conv instantiations of the 8-wave geometry (counting.py's 12 supports, every shape fwd3 does not take)
template <int S, int NOB, bool XVEC, bool MIX, bool EP = false, int NW = 8, bool F16 = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void gml_k_spectconv_fwd2(const GmlFwdParams p) {
    using C = GmlFwd2Cfg<S>;
    using FT = typename GmlPiece<F16>::T;
    constexpr bool HOUT = (NOB == 0);
    static_assert(!F16 || (NOB != 0 && NW == 8), "f16 pieces: conv instantiations, 8-wave geometry");
    static_assert(!(EP && NOB == 0), "the stand-alone SpMM instantiations take contiguous value rows");
    static_assert(NW == 8 || (NW == 4 && NOB != 0), "4-wave geometry: conv instantiations only");
    constexpr int NT = 64 * NW, ECAP = C::ecap(NW), XCAP = C::xcap(NW);
    constexpr int NOBA = HOUT ? 1 : NOB;
    constexpr bool H32 = HOUT && MIX;                          // SpMM instantiations reuse the MIX slot: Fin == 32 (full-line stores)
    constexpr bool MIXB = MIX && !HOUT;
    constexpr bool ROT = HOUT;                                 // loop shape, see below
    // group records one stage ahead of the data they describe: only in the rotated shape.  Measured again in round 2
    // (-DGML_FWD2_REC_AHEAD, tools/ab.sh): with the record prefetched the conv instantiations run 10 % SLOWER (2.34 vs 2.12
    // ms/step), although the dependent record -> data round trip disappears from the issue burst.
#ifdef GML_FWD2_REC_AHEAD
    constexpr bool RECPRE = true;
#else
    constexpr bool RECPRE = ROT;
#endif
    constexpr int LDX = C::LDX, ROWS = C::rows(NW);
    constexpr int VAL_ALIGN = (S % 4 == 0) ? 4 : ((S % 2 == 0) ? 2 : 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* Wof_h = reinterpret_cast<__bf16*>(lds_raw);       // [s][o][f], 16-byte chunks XOR-swizzled by gml_wkey(o)
    __bf16* Wof_l = Wof_h + C::W_HALF;
    int* rp_l = reinterpret_cast<int*>(lds_raw + C::W_BYTES);
    int* col_l = rp_l + ROWS + 8;
    float* ea_l = reinterpret_cast<float*>(col_l + ECAP);
    float* xs = ea_l + ECAP * S;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);
    if (g0 >= g1) return;

    // f16 pieces: the largest magnitude of every output column first (32 words of the column-id area: free until the first commit)
    uint32_t* cmax = reinterpret_cast<uint32_t*>(col_l);
    if constexpr (F16) {
        if (tid < 32) cmax[tid] = 0u;
        __syncthreads();
        {
            const int o = tid & 31;
            float m = 0.f;
            if (o < p.Fout)
                for (int i = tid >> 5; i < S * 32; i += NT / 32) {
                    const int s = i >> 5, f = i & 31;
                    if (f < p.Fin) m = fmaxf(m, fabsf(p.w[(int64_t)(p.s0 + s) * p.w_ss + (int64_t)f * p.w_si + (int64_t)o * p.w_so]));
                }
            atomicMax(&cmax[o], __float_as_uint(m));
        }
        __syncthreads();
    }
    for (int e = tid; e < S * 32 * 32 && !HOUT; e += NT) {
        const int f = e & 31, o = (e >> 5) & 31, s = e >> 10;
        const float v = (f < p.Fin && o < p.Fout) ? p.w[(int64_t)(p.s0 + s) * p.w_ss + (int64_t)f * p.w_si + (int64_t)o * p.w_so] : 0.f;
        const int iof = (s * 32 + o) * 32 + ((((f >> 3) ^ gml_wkey(o)) & 3) << 3) + (f & 7);
        if constexpr (F16) {
            float sc, inv;
            gml_f16_scale_bits(cmax[o], sc, inv);
            const float vs = v * sc;
            const _Float16 h = (_Float16)vs;
            reinterpret_cast<_Float16*>(Wof_h)[iof] = h;
            reinterpret_cast<_Float16*>(Wof_l)[iof] = (_Float16)(vs - (float)h);
        } else {
            const __bf16 h = (__bf16)v;
            const __bf16 l = (__bf16)(v - (float)h);
            Wof_h[iof] = h;
            Wof_l[iof] = l;
        }
    }
    float bias_r[NOBA], winv_r[NOBA];
#pragma unroll
    for (int ob = 0; ob < NOBA; ++ob) {
        bias_r[ob] = (!HOUT && p.bias && ob * 16 + r16 < p.Fout) ? p.bias[ob * 16 + r16] : 0.f;
        winv_r[ob] = 1.f;
        if constexpr (F16) { float sc; gml_f16_scale_bits(cmax[ob * 16 + r16], sc, winv_r[ob]); }
    }

    FT mwh, mwl;                                               // B[k = f][n = c]: c < F2 -> w11 row c, F2 <= c < 2 F2 -> w12 row c - F2
    float mbias = 0.f, mwinv = 1.f;
    if constexpr (MIXB) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = 8 * kq + j;
            const bool ok = f < p.Fin && r16 < 2 * p.F2;
            v[j] = ok ? (r16 < p.F2 ? p.w11[r16 * p.Fin + f] : p.w12[(r16 - p.F2) * p.Fin + f]) : 0.f;
        }
        if constexpr (F16) {
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            float sc;
            gml_f16_scale_bits(__float_as_uint(m), sc, mwinv);
            gml_split8_f16(v, sc, mwh, mwl);
        } else {
            gml_split8(v, mwh, mwl);
        }
        if (r16 < p.F2) mbias = p.b11 ? p.b11[r16] : 0.f;
        else if (r16 < 2 * p.F2) mbias = p.b12 ? p.b12[r16 - p.F2] : 0.f;
    }

    // ---- software-pipelined staging.  Every prefetch load is unconditional with indices clamped into the arrays (lanes
    //      outside fetch a valid, unused element): with a load under a predicate or branch the compiler cannot count the
    //      loads in flight and waits for all of them at the next vmcnt it needs.
    constexpr int NC = ECAP / NT, NE4 = (S % 4 == 0) ? ECAP * (S / 4) / NT : 1;
    constexpr int NX4 = XVEC ? (XCAP * 8 + NT - 1) / NT : 1;
    constexpr int NX1 = XVEC ? 1 : (XCAP * 32 + NT - 1) / NT;
    const int etot = p.rowptr[p.nrows];
    const int* colb = etot > 0 ? p.col : p.ginfo;              // an edgeless graph reads the (always present) group records
    const f32x4* valb = etot > 0 ? reinterpret_cast<const f32x4*>(p.val) : reinterpret_cast<const f32x4*>(p.ginfo);
    const int* eposb = (EP && etot > 0) ? p.epos : p.ginfo;
    const int emax = max(etot, 1) - 1;
    const int64_t emax4 = (S % 4 == 0) ? max((int64_t)etot * (S / 4), (int64_t)1) - 1 : 0;
    static_assert(S % 4 == 0, "float4 value rows");
    constexpr bool vec_ok = true;                              // the dispatcher guarantees p.S == S and 16-byte aligned value rows
    const int f4max = ((p.Fin + 3) / 4 * 4 - 4);
    int cv[NC], rpv = 0;
    int pv[EP ? NE4 : 1];
    f32x4 ev4[NE4], xv4[NX4];
    float xv1[NX1];
    // group records run one stage ahead of the data they describe: rec_* = record of the group whose data loads are
    // issued next (its kb / lo are load addresses, and waiting for a record loaded right there would also wait, in
    // order, for every store issued before it)
    int4 rec_gi = int4{0, 0, 0, 0};
    int rec_row = 0;
    uint32_t rec_outrows = 0;
    int4 gi_n = int4{0, 0, 0, 0};                              // record of the group whose data is in the registers
    int row_n = 0;
    uint32_t outrows_n = 0;
    auto load_rec = [&](int g) {
        const int32_t* rec = p.ginfo + (int64_t)g * GML_GREC_INTS(NW == 8 ? 128 : GML_GROUPS64_RANKED);
        rec_gi = *reinterpret_cast<const int4*>(rec);
        rec_row = reinterpret_cast<const unsigned char*>(rec + 4)[wave * 16 + r16];
        rec_outrows = reinterpret_cast<const uint32_t*>(rec + 4)[wave * 4 + kq];
    };
    auto issue = [&](int g, int gnext) {                       // data loads of group g (record in rec_*), record of gnext
        if constexpr (!RECPRE) load_rec(g);                    // (record and data in one stage: a dependent round trip per group)
        gi_n = rec_gi; row_n = rec_row; outrows_n = rec_outrows;
        const int64_t r0 = (int64_t)g * ROWS;
        rpv = p.rowptr[min(r0 + tid, p.nrows)];
        const int kb = gi_n.x, lo = gi_n.z;
        if constexpr (EP) {                                    // first in the burst: they gate issue_vals()
#pragma unroll
            for (int t = 0; t < NE4; ++t) pv[t] = eposb[min(kb + (tid + NT * t) / (S / 4), emax)];
        }
#pragma unroll
        for (int t = 0; t < NC; ++t) cv[t] = colb[min(kb + tid + NT * t, emax)];
        if constexpr (!EP) {
#pragma unroll
            for (int t = 0; t < NE4; ++t) ev4[t] = valb[min((int64_t)kb * (S / 4) + tid + NT * t, emax4)];
        }
        if constexpr (XVEC) {
#pragma unroll
            for (int t = 0; t < NX4; ++t) {
                const int i = tid + NT * t;
                const int64_t rr = min((int64_t)lo + (i >> 3), p.nrows - 1);
                xv4[t] = *reinterpret_cast<const f32x4*>(p.x + rr * p.ldx + min((i & 7) * 4, f4max));
            }
        } else {
#pragma unroll
            for (int t = 0; t < NX1; ++t) {
                const int i = tid + NT * t;
                const int64_t rr = min((int64_t)lo + (i >> 5), p.nrows - 1);
                xv1[t] = p.x[rr * p.ldx + min(i & 31, p.Fin - 1)];
            }
        }
        if constexpr (RECPRE) load_rec(gnext);                 // last in the burst: nothing of this stage waits for it
    };
    auto issue_vals = [&]() {                                  // EP: the value rows at the positions that arrived meanwhile
#pragma unroll
        for (int t = 0; t < NE4; ++t)
            ev4[t] = valb[min((int64_t)(EP ? pv[t] : 0) * (S / 4) + ((tid + NT * t) % (S / 4)), emax4)];
    };
    // the staged group's description (latched by commit)
    int kb = 0, ne = 0, lo = 0, row = 0;
    uint32_t out_rows = 0;
    bool staged = false;
    auto commit = [&](int g) {                                 // prefetched registers of group g -> LDS
        const int nr = (int)min((int64_t)ROWS, p.nrows - (int64_t)g * ROWS);
        kb = gi_n.x; ne = gi_n.y; lo = gi_n.z;
        const int nwin = gi_n.w;
        row = row_n; out_rows = outrows_n;
        staged = vec_ok && ne <= ECAP && nwin <= XCAP;
        if (tid <= nr) rp_l[tid] = rpv;
        if (staged) {
#pragma unroll
            for (int t = 0; t < NC; ++t) { const int i = tid + NT * t; if (i < ne) col_l[i] = cv[t] - lo; }
#pragma unroll
            for (int t = 0; t < NE4; ++t) {
                const int i = tid + NT * t;
                if (i < ne * (S / 4)) reinterpret_cast<f32x4*>(ea_l)[i] = ev4[t];
            }
            if constexpr (XVEC) {
#pragma unroll
                for (int t = 0; t < NX4; ++t) {
                    const int i = tid + NT * t;
                    const int f4 = (i & 7) * 4;
                    if (i < nwin * 8)
                        *reinterpret_cast<f32x4*>(xs + (i >> 3) * LDX + f4) = (f4 < p.Fin) ? xv4[t] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            } else {
#pragma unroll
                for (int t = 0; t < NX1; ++t) {
                    const int i = tid + NT * t;
                    if (i < nwin * 32) xs[(i >> 5) * LDX + (i & 31)] = ((i & 31) < p.Fin) ? xv1[t] : 0.f;
                }
            }
        }
    };
    // Loop shape (ROT): issue(g + 1) -> compute(g) -> stores(g) -> barrier -> commit(g + 1) -> barrier.  The commit's
    // wait sits in the same straight-line region as the loads it waits for and the (unpredicated) stores issued after
    // them, so it is an exact vmcnt for the loads; with the commit at the loop top it merges with the store-less entry
    // path into vmcnt(0) and every group waits for the previous group's stores to drain.  On the last trip the
    // prefetch re-reads the same group (cache hits, keeps the counts static) and its commit is skipped.
    // Only the SpMM instantiations (store-bound) are rotated: the conv instantiations measured 4-5 % FASTER with the
    // commit at the loop top and the record fetched with its data (their projection phase already stalls on the
    // prefetched registers it has to recycle), the SpMM 2 % slower.
    if constexpr (RECPRE) load_rec(g0);
    issue(g0, min(g0 + 1, g1 - 1));
    if constexpr (EP) issue_vals();
    if constexpr (ROT) commit(g0);
    __syncthreads();                                           // W images (and, rotated, the first group's staging) complete

#ifdef GML_FWD2_TIMING
    unsigned tacc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tprev_ = (unsigned)__builtin_readcyclecounter();
#endif
    for (int g = g0; g < g1; ++g) {
        const int64_t r0 = (int64_t)g * ROWS;
        const int nr = (int)min((int64_t)ROWS, p.nrows - r0);
        const int gn = min(g + 1, g1 - 1);
        if constexpr (ROT) {
            issue(gn, min(g + 2, g1 - 1));                     // in flight during this group's compute
        } else {
            commit(g);
            GML_TF(0);
            __syncthreads();
            GML_TF(1);
            if (g + 1 < g1) issue((GML_FWABL & 8) ? g0 : g + 1, min(g + 2, g1 - 1));
            GML_TF(2);
        }
        const bool rvalid = row < nr;
        float xrow[MIXB ? 8 : 1];                              // the lane's own x row, features 8*kq..8*kq+7 (Hadamard branch)
        if constexpr (MIXB) {
            const float* xr = p.x + min(r0 + row, p.nrows - 1) * p.ldx;
            if constexpr (XVEC) {
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(xr + min(8 * kq + 4 * q4, f4max));
                    xrow[4 * q4] = t.x; xrow[4 * q4 + 1] = t.y; xrow[4 * q4 + 2] = t.z; xrow[4 * q4 + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xrow[j] = xr[min(8 * kq + j, p.Fin - 1)];
            }
        }

        const int kbeg = rvalid ? rp_l[row] : 0;
        const int kend = rvalid ? rp_l[row + 1] : 0;
        GML_TF(3);

        // ---- aggregation (fp32 VALU, packed): acc[s][f] += val[k, s] * x[col[k], f], f = 8*kq .. 8*kq+7
        f32x2 acc[S][4];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int h = 0; h < 4; ++h) acc[s][h] = f32x2{0.f, 0.f};
        if (staged) {
            for (int k = kbeg - kb; k < ((GML_FWABL & 1) ? kbeg - kb : kend - kb); ++k) {
                const int srcl = col_l[k];
                float ev[S];
                gml_load_row<S, VAL_ALIGN>(ea_l + k * S, ev);
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(xs + srcl * LDX + 8 * kq);
                const f32x4 t1 = *reinterpret_cast<const f32x4*>(xs + srcl * LDX + 8 * kq + 4);
                const f32x2 xv[4] = {f32x2{t0.x, t0.y}, f32x2{t0.z, t0.w}, f32x2{t1.x, t1.y}, f32x2{t1.z, t1.w}};
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const f32x2 e2 = f32x2{ev[s], ev[s]};
#pragma unroll
                    for (int h = 0; h < 4; ++h) acc[s][h] = e2 * xv[h] + acc[s][h];
                }
            }
        } else {                                               // group outside the LDS capacities (or unaligned rows): global gathers
            for (int k = kbeg; k < kend; ++k) {
                const int src = p.col[k];
                float xb[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) xb[t] = (8 * kq + t < p.Fin) ? p.x[(int64_t)src * p.ldx + 8 * kq + t] : 0.f;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float e = p.val[(int64_t)(EP ? p.epos[k] : k) * p.S + p.s0 + s];
                    const f32x2 e2 = f32x2{e, e};
#pragma unroll
                    for (int h = 0; h < 4; ++h) acc[s][h] = e2 * f32x2{xb[2 * h], xb[2 * h + 1]} + acc[s][h];
                }
            }
        }
        GML_TF(4);
        if constexpr (EP) {
            if (g + 1 < g1) issue_vals();                      // next group's value rows (their positions were issued at the top)
        }
        GML_TF(5);

        if constexpr (HOUT) {                                  // stand-alone SpMM: the aggregate is the output
            // H[row][s][0..Fin) is 4 * Fin bytes: a lane's 8 features are a quarter of it.  Through a per-wave LDS tile
            // ([16 rows][36], in the otherwise idle W area) every store instruction writes whole 128-byte (row, s)
            // segments: lane l -> segment l >> 3, 16 bytes each.  The stores go through a buffer descriptor of exactly
            // this group's rows: a lane whose tile row lies beyond the last row is dropped by the hardware range check,
            // so the store stream has no predicate and the compiler can count it -- the next group's staging waits
            // for its loads only, not (vmcnt(0)) for these stores to drain.  H is written once and is far larger than
            // the caches: non-temporal stores (measured 187 -> 175 us on the 769 MB of the ZINC batch).
            float* tile = reinterpret_cast<float*>(lds_raw) + wave * (16 * 36);
            if constexpr (H32) {
                const int64_t rowb = (int64_t)S * 32 * 4;
                float* gbase = p.hout + r0 * ((int64_t)S * 32);
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, (int)(nr * rowb), 0x00020000);
                int voff[2];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int rr = __shfl(row, half * 8 + (lane >> 3));                   // lane `pos` of this wave owns tile row pos
                    voff[half] = (rr < nr) ? rr * (int)rowb + (lane & 7) * 16 : 0x7fffff00;
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    *reinterpret_cast<f32x4*>(tile + r16 * 36 + 8 * kq) = f32x4{acc[s][0].x, acc[s][0].y, acc[s][1].x, acc[s][1].y};
                    *reinterpret_cast<f32x4*>(tile + r16 * 36 + 8 * kq + 4) = f32x4{acc[s][2].x, acc[s][2].y, acc[s][3].x, acc[s][3].y};
                    // (LDS ops of one wave execute in order: no barrier inside the wave's private tile)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int pos = half * 8 + (lane >> 3), c4 = (lane & 7) * 4;      // tile position -> its row
                        const u32x4 t = *reinterpret_cast<const u32x4*>(tile + pos * 36 + c4);
                        __builtin_amdgcn_raw_buffer_store_b128(t, rs, voff[half], s * 128, /*nt*/ 2);
                    }
                }
            } else {                                           // Fin < 32: element stores, same unpredicated form
                const int rowb = S * p.Fin * 4;
                float* gbase = p.hout + r0 * ((int64_t)S * p.Fin);
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, nr * rowb, 0x00020000);
                int voff[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    voff[j] = (rvalid && 8 * kq + j < p.Fin) ? row * rowb + (8 * kq + j) * 4 : 0x7fffff00;
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[s][h].x), rs, voff[2 * h], s * p.Fin * 4, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[s][h].y), rs, voff[2 * h + 1], s * p.Fin * 4, 0);
                    }
            }
            __syncthreads();                                   // this group's LDS reads are done
            if constexpr (ROT) {
                if (g + 1 < g1) commit(gn);
                __syncthreads();
            }
            continue;
        }

        // ---- projection: out tile = sum_s acc_s W_s (acc split on the fly = A fragments, k = f = 8*kq + j)
        f32x4 oacc[NOBA], oold[F16 ? NOBA : 1];
        float oscale[NOBA];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) { oacc[ob] = f32x4{0.f, 0.f, 0.f, 0.f}; oscale[ob] = 1.f; }
        if constexpr (F16) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) oold[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (p.flags & GML_ACCUM) {                             // the old values travel while the MFMAs run (clamped loads)
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int lr = min((int)((out_rows >> (8 * reg)) & 255u), nr - 1);
                    const float ov = p.out[(r0 + lr) * p.ldo + min(ob * 16 + r16, p.Fout - 1)];
                    if constexpr (F16) oold[ob][reg] = ov;     // (the accumulators are in scaled units: added in the epilogue)
                    else oacc[ob][reg] = ov;
                }
        }
        {
            // W fragments of support s + 1 are requested before the MFMAs of support s (the projection as first written --
            // read, wait, dependent MFMA triple, per block -- exposed one LDS round trip per block)
            FT wh[2][NOBA], wl[2][NOBA];
            auto frag = [&](int s, int st) {
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) {
                    const int o = ob * 16 + r16;               // B[k = f][n = o]: 8 consecutive f of column o
                    const int off = (s * 32 + o) * 32 + (((kq ^ gml_wkey(o)) & 3) << 3);
                    wh[st][ob] = *reinterpret_cast<const FT*>(Wof_h + off);
                    wl[st][ob] = *reinterpret_cast<const FT*>(Wof_l + off);
                }
            };
            frag(0, 0);
            float asc = 1.f;
            if constexpr (F16) {                               // the tile's scale (gml_spectconv_fwd3_impl.h)
                float m = 0.f;
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int h = 0; h < 4; ++h) m = fmaxf(fmaxf(fabsf(acc[s][h].x), fabsf(acc[s][h].y)), m);
                float ainv;
                gml_f16_scale_bits(gml_wave_max_bits(__float_as_uint(m)), asc, ainv);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oscale[ob] = ainv * winv_r[ob];
            }
#pragma unroll
            for (int s = 0; s < ((GML_FWABL & 2) ? 0 : S); ++s) {
                const int st = s & 1;
                if (s + 1 < S) frag(s + 1, st ^ 1);
                const float av[8] = {acc[s][0].x, acc[s][0].y, acc[s][1].x, acc[s][1].y, acc[s][2].x, acc[s][2].y, acc[s][3].x, acc[s][3].y};
                FT ah, al;
                if constexpr (F16) gml_split8_f16(av, asc, ah, al);
                else gml_split8(av, ah, al);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oacc[ob] = gml_mfma_piece(al, wh[st][ob], oacc[ob]);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oacc[ob] = gml_mfma_piece(ah, wl[st][ob], oacc[ob]);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oacc[ob] = gml_mfma_piece(ah, wh[st][ob], oacc[ob]);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NOB, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NOB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NOB, 0);
            }
        }
        GML_TF(6);
        // output stores through a buffer descriptor based at this group's first row: lanes outside (row >= nr,
        // column >= Fout) get an offset beyond the range and are dropped by the hardware -- no predicate, so the
        // compiler counts the stores and the next group's commit does not wait for them (see the SpMM branch above)
        const auto ors = __builtin_amdgcn_make_buffer_rsrc(p.out + r0 * p.ldo, 0, 0x7ffffe00, 0x00020000);
        const bool relu = (p.flags & GML_RELU) != 0;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const int o = ob * 16 + r16;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int lr = (int)((out_rows >> (8 * reg)) & 255u);
                float v = F16 ? fmaf(oacc[ob][reg], oscale[ob], bias_r[ob]) + oold[F16 ? ob : 0][reg] : oacc[ob][reg] + bias_r[ob];
                if (relu) v = fmaxf(v, 0.f);
                const int off = (o < p.Fout && lr < nr && !(GML_FWABL & 4)) ? (lr * (int)p.ldo + o) * 4 : 0x7fffff00;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, off, 0, 0);
            }
        }
        GML_TF(7);
        if constexpr (MIXB) {
            // z[row][c] = x[row] . wmix[c]: A = the lane's row (k = f), D: lane (c = r16, kq) holds rows 4*kq + reg
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (8 * kq + j >= p.Fin) xrow[j] = 0.f;        // clamped loads fetched a neighbour: outside Fin -> 0
            FT xh, xl;
            float zsc = 1.f;
            if constexpr (F16) {
                float m = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(xrow[j]));
                float xsc, xinv;
                gml_f16_scale_bits(gml_wave_max_bits(__float_as_uint(m)), xsc, xinv);
                gml_split8_f16(xrow, xsc, xh, xl);
                zsc = xinv * mwinv;
            } else {
                gml_split8(xrow, xh, xl);
            }
            f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            z = gml_mfma_piece(xl, mwh, z);
            z = gml_mfma_piece(xh, mwl, z);
            z = gml_mfma_piece(xh, mwh, z);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float t = gml_tanh(fmaf(z[reg], zsc, mbias));
                const float u = __shfl(t, lane + p.F2);        // partner column c + F2 of the same 16-lane row group
                const int lr = (int)((out_rows >> (8 * reg)) & 255u);
                const int off = (r16 < p.F2 && lr < nr) ? (lr * (int)p.ldo + p.mix_col + r16) * 4 : 0x7fffff00;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t * u), ors, off, 0, 0);
            }
        }
        GML_TF(8);
        __syncthreads();                                       // this group's LDS reads are done
        GML_TF(9);
        if constexpr (ROT) {
            if (g + 1 < g1) commit(gn);
            __syncthreads();
        }
    }
#ifdef GML_FWD2_TIMING
    if (lane == 0 && p.prof != nullptr) {
#pragma unroll
        for (int i = 0; i < 10; ++i) atomicAdd(&p.prof[i], (unsigned long long)tacc_[i]);
    }
#endif
}

template <int S, int NOB>
int gml_launch_fwd2(const GmlFwdParams& p, dim3 grid, hipStream_t st, bool xvec, bool mix);

#define GML_FWD2_LAUNCH_N(SV, NOBV, XV, MX, EPV, NWV)                                                        \
    {                                                                                                        \
        GML_ALLOW_BIG_LDS(rc_, (&gml_k_spectconv_fwd2<SV, NOBV, XV, MX, EPV, NWV>), 160 * 1024)              \
        if (rc_ != hipSuccess) return (int)rc_;                                                              \
        hipLaunchKernelGGL((gml_k_spectconv_fwd2<SV, NOBV, XV, MX, EPV, NWV>), grid, dim3(64 * NWV),         \
                           GmlFwd2Cfg<SV>::lds_bytes(NWV), st, p);                                           \
        return gml_launch_status();                                                                          \
    }
#define GML_FWD2_LAUNCH_F(SV, NOBV, XV, MX, EPV)                                                             \
    {                                                                                                        \
        GML_ALLOW_BIG_LDS(rc_, (&gml_k_spectconv_fwd2<SV, NOBV, XV, MX, EPV, 8, true>), 160 * 1024)          \
        if (rc_ != hipSuccess) return (int)rc_;                                                              \
        hipLaunchKernelGGL((gml_k_spectconv_fwd2<SV, NOBV, XV, MX, EPV, 8, true>), grid, dim3(512),          \
                           GmlFwd2Cfg<SV>::lds_bytes(8), st, p);                                             \
        return gml_launch_status();                                                                          \
    }
#define GML_FWD2_LAUNCH_E(SV, NOBV, XV, MX, EPV)                                                             \
    {                                                                                                        \
        if constexpr (NOBV != 0) { if (p.flags & GML_F16X3) GML_FWD2_LAUNCH_F(SV, (NOBV != 0 ? NOBV : 1), XV, MX, EPV) }  \
        GML_FWD2_LAUNCH_N(SV, NOBV, XV, MX, EPV, 8)                                                          \
    }
#define GML_FWD2_LAUNCH(SV, NOBV, XV, MX) GML_FWD2_LAUNCH_E(SV, NOBV, XV, MX, false)
#define GML_DEFINE_SPMM2(SV)                                                                                 \
    template <>                                                                                              \
    int gml_launch_fwd2<SV, 0>(const GmlFwdParams& p, dim3 grid, hipStream_t st, bool xvec, bool) {          \
        if (xvec && p.Fin == 32) GML_FWD2_LAUNCH(SV, 0, true, true)                                          \
        if (xvec) GML_FWD2_LAUNCH(SV, 0, true, false)                                                        \
        GML_FWD2_LAUNCH(SV, 0, false, false)                                                                 \
    }
#define GML_DEFINE_FWD2(SV, NOBV)                                                                            \
    template <>                                                                                              \
    int gml_launch_fwd2<SV, NOBV>(const GmlFwdParams& p, dim3 grid, hipStream_t st, bool xvec, bool mix) {   \
        if (p.nw == 4) {   /* 4-wave geometry: float4-addressable x only (the dispatcher checks) */          \
            if (!xvec) return GML_E_UNSUPPORTED;                                                             \
            if (p.epos != nullptr && mix) GML_FWD2_LAUNCH_N(SV, NOBV, true, true, true, 4)                   \
            if (p.epos != nullptr) GML_FWD2_LAUNCH_N(SV, NOBV, true, false, true, 4)                         \
            if (mix) GML_FWD2_LAUNCH_N(SV, NOBV, true, true, false, 4)                                       \
            GML_FWD2_LAUNCH_N(SV, NOBV, true, false, false, 4)                                               \
        }                                                                                                    \
        if (p.epos != nullptr) {                                                                             \
            if (xvec && mix) GML_FWD2_LAUNCH_E(SV, NOBV, true, true, true)                                   \
            if (xvec) GML_FWD2_LAUNCH_E(SV, NOBV, true, false, true)                                         \
            if (mix) GML_FWD2_LAUNCH_E(SV, NOBV, false, true, true)                                          \
            GML_FWD2_LAUNCH_E(SV, NOBV, false, false, true)                                                  \
        }                                                                                                    \
        if (xvec && mix) GML_FWD2_LAUNCH(SV, NOBV, true, true)                                               \
        if (xvec) GML_FWD2_LAUNCH(SV, NOBV, true, false)                                                     \
        if (mix) GML_FWD2_LAUNCH(SV, NOBV, false, true)                                                      \
        GML_FWD2_LAUNCH(SV, NOBV, false, false)                                                              \
    }
