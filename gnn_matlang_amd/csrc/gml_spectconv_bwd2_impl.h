// Fused backward, bf16x3 variant (default for Fin <= 32, 16 < Fout <= 32): same outputs as
// gml_k_spectconv_bwd (gml_spectconv_bwd_impl.h)
//
//   dX = sum_s A_s (G W_s^T),   dval[e,s] = < X[src] W_s, G[dst] >,   dW_s = X^T (A_s G)
//
// but all three projections run on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16) with both operands split
// into bf16 (hi, lo) pairs and the three significant products accumulated in fp32 (fp32-class, see
// gml_common.h).  Why: the f32-input MFMA executes on the fp32 vector ALUs, so in the f32 kernel the
// 3 x 128 MFMAs of a tile serialise with its edge phase; the bf16 pipe is separate and asynchronous.
//
// Workgroup = 8 waves, group = 128 source rows (8 tiles of 16).  LDS: W twice as bf16 (hi, lo), once
// f-contiguous [s][o][f] (A fragments of Z^T = W^T X^T) and once o-contiguous [s][f][o] (B fragments of
// dX = P W^T) -- a bf16 MFMA fragment is 8 consecutive k per lane, so each contraction needs its own
// layout; staging as in the f32 kernel.  Lane (r16, kq) of a tile owns row r16 and the 8 outputs
// o = 8*kq .. 8*kq+7.  dW contracts over the 128 rows: X^T and P^T go through LDS as bf16 (hi, lo)
// [f or o][row]; two supports per slab, wave w owns block (se, fb, ob) = w of the slab.
#pragma once
#include "gml_common.h"
#include "gml_spectconv_bwd_impl.h"

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#define GML_BWD2_ROWS 128
#define GML_BWD2_ECAP_MAX 1024     // register-batched staging bounds (per 128-row group)
#define GML_BWD2_XCAP_MAX 224

template <int S, int NFB>
struct GmlBwd2Cfg {
    static constexpr int FINP = 32, FOUTP = 32;             // K of one bf16 MFMA; NFB = live 16-wide Fin blocks
    static constexpr int LDG = FOUTP + 4;                    // G window rows (floats, b128 aligned)
    static constexpr int W_HALF = S * 32 * 32;               // bf16 elements of one (hi or lo) W image
    static constexpr int W_BYTES = 4 * W_HALF * 2;           // fo-hi, fo-lo, of-hi, of-lo
    static constexpr int SE = 2;                             // supports per dW slab (8 blocks -> 8 waves when NFB=2)
    static constexpr int NSLAB = S / SE;
    static constexpr int LDT = GML_BWD2_ROWS + 16;          // row stride (bf16) of the transposed tiles; with the 16-byte
                                                             // chunks of every 64-byte K step XOR-ed by (channel >> 2) & 3
                                                             // the fragment reads are conflict-free (tools/lds_sim.py)
    static constexpr int XT_BYTES = 2 * 32 * LDT * 2;                  // X^T hi, lo  [f][row]
    static constexpr int PT_BYTES = 2 * SE * 32 * LDT * 2;             // P^T hi, lo  [se][o][row]
    static constexpr bool OK = (S % SE == 0);
    __host__ __device__ static size_t ea_bytes(int ecap) {
        const size_t a = (size_t)ecap * S * 4;
        return a > (size_t)PT_BYTES ? a : (size_t)PT_BYTES;   // the value rows' region later holds the P^T slab
    }
    __host__ __device__ static size_t gs_bytes(int xcap) {
        const size_t a = (size_t)xcap * LDG * 4;
        return a > (size_t)XT_BYTES ? a : (size_t)XT_BYTES;   // the G window's region later holds X^T
    }
    __host__ __device__ static size_t lds_bytes(int ecap, int xcap) {
        return (size_t)W_BYTES + 136 * 4 + (size_t)ecap * 4 + ea_bytes(ecap) + gs_bytes(xcap);
    }
};

#ifdef GML_BWD2_TIMING      // debug build: per-phase cycle sums of thread 0 of every workgroup (tools/bwd2_phases.py)
#define GML_T(i) do { if (tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
                      atomicAdd(&p.prof[i], t_ - tprev_); tprev_ = t_; } } while (0)
#else
#define GML_T(i)
#endif

// XV: the X rows are float4-addressable (p.xvec); a template parameter so that each instantiation carries one load path
template <int S, int NFB, bool XV>
__global__ __launch_bounds__(512) void gml_k_spectconv_bwd2(const GmlBwdParams p) {
    using C = GmlBwd2Cfg<S, NFB>;
    constexpr int LDG = C::LDG;
    constexpr int ROWS = GML_BWD2_ROWS, LDT = C::LDT;
    constexpr int VAL_ALIGN = (S % 4 == 0) ? 4 : ((S % 2 == 0) ? 2 : 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* Wfo_h = reinterpret_cast<__bf16*>(lds_raw);     // [s][f][o]
    __bf16* Wfo_l = Wfo_h + C::W_HALF;
    __bf16* Wof_h = Wfo_l + C::W_HALF;                       // [s][o][f]
    __bf16* Wof_l = Wof_h + C::W_HALF;
    int* rp_l = reinterpret_cast<int*>(lds_raw + C::W_BYTES);
    int* col_l = rp_l + 136;
    float* ea_l = reinterpret_cast<float*>(col_l + p.ecap);
    float* gs = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ea_l) + C::ea_bytes(p.ecap));
    __bf16* pT_h = reinterpret_cast<__bf16*>(ea_l);          // [se][o][row]   (after the dval rows left)
    __bf16* pT_l = pT_h + C::SE * 32 * LDT;
    __bf16* xT_h = reinterpret_cast<__bf16*>(gs);            // [f][row]        (after the edge phase)
    __bf16* xT_l = xT_h + 32 * LDT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);

    // W -> bf16 (hi, lo), both layouts, zero padded to 32 x 32
    for (int e = tid; e < S * 32 * 32; e += 512) {
        const int o = e & 31, f = (e >> 5) & 31, s = e >> 10;
        const float v = (f < p.Fin && o < p.Fout) ? p.w[((int64_t)s * p.Fin + f) * p.Fout + o] : 0.f;
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        // 16-byte chunks of a 64-byte row are XOR-swizzled (gml_wkey) so that the rows one ds_read_b128 lane group
        // touches land on distinct banks
        const int ifo = (s * 32 + f) * 32 + ((((o >> 3) ^ gml_wkey(f)) & 3) << 3) + (o & 7);
        const int iof = (s * 32 + o) * 32 + ((((f >> 3) ^ gml_wkey(o)) & 3) << 3) + (f & 7);
        Wfo_h[ifo] = h; Wfo_l[ifo] = l;
        Wof_h[iof] = h; Wof_l[iof] = l;
    }

    f32x4 dwacc[C::NSLAB];
#pragma unroll
    for (int i = 0; i < C::NSLAB; ++i) dwacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef GML_BWD2_TIMING
    unsigned long long tprev_ = __builtin_readcyclecounter();
#endif
    // ---- staging registers.  A group's global loads are issued one phase early -- after the PREVIOUS group's dX
    //      stores, before its dW phase (which touches no global memory and leaves the registers of the edge / dX phases
    //      free) -- and committed to LDS at the top of the group, so the load latency runs under the dW arithmetic.
    constexpr int NC = GML_BWD2_ECAP_MAX / 512, NE4 = (S % 4 == 0) ? GML_BWD2_ECAP_MAX * (S / 4) / 512 : 1;
    constexpr int NG4 = (GML_BWD2_XCAP_MAX * 8 + 511) / 512;
    const int etot = p.rowptr[p.nrows];
    const int* colb = etot > 0 ? p.col : p.ginfo;              // an edgeless graph reads the (always present) group records
    const f32x4* valb = etot > 0 ? reinterpret_cast<const f32x4*>(p.val) : reinterpret_cast<const f32x4*>(p.ginfo);
    const int emax = max(etot, 1) - 1;
    const int64_t emax4 = (S % 4 == 0) ? max((int64_t)etot * (S / 4), (int64_t)1) - 1 : 0;
    const int o4max = p.gvec ? ((p.Fout + 3) / 4 * 4 - 4) : 0;   // float4 rows of g only when gvec (else unused anyway)
    int cv[NC], rpv = 0, row_n = 0;
    uint32_t outrows_n = 0;
    f32x4 ev4[NE4], gv4[NG4];
    float xb[8];
    auto vec_group = [&](const int4 gi) {
        return (S % 4 == 0) && p.gvec && gi.y <= GML_BWD2_ECAP_MAX && gi.w <= GML_BWD2_XCAP_MAX;
    };
    // {kb, ne, lo, nwin} of the group whose loads are issued next (fetched with its row bytes a group ahead: it holds
    // load ADDRESSES, and a record fetched right there would stall the issue for a full latency) and of the group
    // being processed (wave-uniform: kept in scalar registers)
    int4 gi_nv = int4{0, 0, 0, 0};
    int4 gi_c = int4{0, 0, 0, 0};
    // latch(): the record fetched a group ahead -> scalar registers.  Called BEFORE the old-dx loads of the phase: its
    // wait (the record's vector load, counted conservatively) must not sit behind loads issued just before it.
    auto latch = [&]() {
        gi_c = int4{__builtin_amdgcn_readfirstlane(gi_nv.x), __builtin_amdgcn_readfirstlane(gi_nv.y),
                    __builtin_amdgcn_readfirstlane(gi_nv.z), __builtin_amdgcn_readfirstlane(gi_nv.w)};
    };
    auto issue = [&](int g) {
        const int64_t r0 = (int64_t)g * ROWS;
        const int nr = (int)min((int64_t)ROWS, p.nrows - r0);
        const int4 gi = gi_c;
        const int kb = gi.x, ne = gi.y, lo = gi.z, nwin = gi.w;
        // the lane's own x row (row_n: loaded a whole group earlier, see load_rows): unconditional, clamped loads --
        // a select on the loaded value here would wait for it on the spot; rows / features outside are zeroed when
        // the registers are used (group top)
        if constexpr (XV) {
            const float* xr = p.x + min(r0 + row_n, p.nrows - 1) * p.ldx;
            const int f4max = (p.Fin + 3) / 4 * 4 - 4;
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(xr + min(8 * kq + 4 * q4, f4max));
                xb[4 * q4] = t.x; xb[4 * q4 + 1] = t.y; xb[4 * q4 + 2] = t.z; xb[4 * q4 + 3] = t.w;
            }
        }                                                     // (rows that are not float4-addressable: loaded at the group top)
        (void)nr;
        rpv = p.rowptr[min(r0 + tid, p.nrows)];
        // Unconditional loads, indices clamped into the arrays (lanes outside fetch a valid, unused element): the
        // compiler can count them, so the dX phase's wait for the old dx values (issued just before these) is an
        // exact vmcnt and does not cover them.  (A common branch around them would also keep the registers' old
        // values live through the whole trip.)
        // (clamped to the group's OWN last element, not the array's: the register batches cover the largest group
        //  the path takes, and a lane beyond this group must not drag the following groups' data in)
        const int ne1 = max(ne, 1) - 1, ne41 = max(ne * (S / 4), 1) - 1, nw1 = max(nwin, 1) - 1;
#pragma unroll
        for (int t = 0; t < NC; ++t) cv[t] = colb[min(kb + min(tid + 512 * t, ne1), emax)];
#pragma unroll
        for (int t = 0; t < NE4; ++t) ev4[t] = valb[min((int64_t)kb * (S / 4) + min(tid + 512 * t, ne41), emax4)];
#pragma unroll
        for (int t = 0; t < NG4; ++t) {
            const int i = tid + 512 * t;
            const int64_t rr = min((int64_t)lo + min(i >> 3, nw1), p.nrows - 1);
            gv4[t] = *reinterpret_cast<const f32x4*>(p.g + rr * p.ldg + min((i & 7) * 4, o4max));
        }
    };
    // degree-sorted row assignment: this lane's row, and the rows of its 4 dX output registers (one byte each)
    auto load_rows = [&](int g) {
        const int32_t* rec = p.ginfo + (int64_t)g * GML_GREC_INTS(ROWS);
        gi_nv = *reinterpret_cast<const int4*>(rec);
        row_n = reinterpret_cast<const unsigned char*>(rec + 4)[wave * 16 + r16];
        outrows_n = reinterpret_cast<const uint32_t*>(rec + 4)[wave * 4 + kq];
    };
    if (g0 < g1) { load_rows(g0); latch(); issue(g0); }
    for (int g = g0; g < g1; ++g) {
        const int64_t r0 = (int64_t)g * ROWS;
        const int nr = (int)min((int64_t)ROWS, p.nrows - r0);
        const int4 gi = gi_c;                                // latched when this group's loads were issued
        const int kb = gi.x, ne = gi.y, lo = gi.z, nwin = gi.w;
        const int row = row_n;
        const uint32_t out_rows = outrows_n;
        load_rows(min(g + 1, g1 - 1));                       // next group's rows: needed as addresses when its loads are issued
        __syncthreads();                                     // previous group is done with every LDS region
        GML_T(0);                                            // = previous group's dW phase + this barrier

        // ---- stage: commit the registers loaded one phase ago
        const bool rvalid = row < nr;
        if constexpr (!XV) {                                 // element-wise x row: issued here, used after the commit
            const float* xr = p.x + (r0 + row) * p.ldx + 8 * kq;
#pragma unroll
            for (int t = 0; t < 8; ++t) xb[t] = (rvalid && 8 * kq + t < p.Fin) ? xr[t] : 0.f;
        }
        if (tid <= nr) rp_l[tid] = rpv;
        if (vec_group(gi)) {
#pragma unroll
            for (int t = 0; t < NC; ++t) { const int i = tid + 512 * t; if (i < ne) col_l[i] = cv[t] - lo; }
#pragma unroll
            for (int t = 0; t < NE4; ++t) {
                const int i = tid + 512 * t;
                if (i < ne * (S / 4)) reinterpret_cast<f32x4*>(ea_l)[i] = ev4[t];
            }
#pragma unroll
            for (int t = 0; t < NG4; ++t) {
                const int i = tid + 512 * t;
                if (i < nwin * 8)
                    *reinterpret_cast<f32x4*>(gs + (i >> 3) * LDG + (i & 7) * 4) = ((i & 7) * 4 < p.Fout) ? gv4[t] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
            for (int i = tid; i < ne; i += 512) col_l[i] = p.col[kb + i] - lo;
            for (int i = tid; i < ne * S; i += 512) ea_l[i] = p.val[(int64_t)kb * S + i];
            for (int i = tid; i < nwin * 32; i += 512) {
                const int rr = i >> 5, o = i & 31;
                gs[rr * LDG + o] = (o < p.Fout) ? p.g[(int64_t)(lo + rr) * p.ldg + o] : 0.f;
            }
        }
        __syncthreads();
        GML_T(1);
        // old dx values (accumulate mode): loaded here, a whole Z + edge phase before the dX chain adds onto them
        f32x4 dxa[NFB];
        {
            const float* dxb = p.dx ? p.dx : p.x;            // (no dx wanted: any readable rows, the values are dropped)
            const int64_t ldb = p.dx ? p.lddx : p.ldx;
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) {
                const int f = fb * 16 + r16;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int lr = min((int)((out_rows >> (8 * reg)) & 255u), nr - 1);
                    dxa[fb][reg] = dxb[(r0 + lr) * ldb + min(f, p.Fin - 1)];
                }
            }
        }

        const int kbeg = rvalid ? rp_l[row] - kb : 0;
        const int kend = rvalid ? rp_l[row + 1] - kb : 0;

        // own X row (its load was issued ahead of the staging loads), features 8*kq .. 8*kq+7  ->  bf16 (hi, lo)
        // B fragment of Z^T, also the X^T tile of dW
        bf16x8 xh, xl;
        if constexpr (XV) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!(rvalid && 8 * kq + j < p.Fin)) xb[j] = 0.f;    // the prefetch loaded clamped addresses
        }
        gml_split8(xb, xh, xl);

        // ---- Z^T = W^T X^T: MFMA row i of block ob carries o = 8*(i>>2) + 4*ob + (i&3), so lane kq receives
        //      o = 8*kq + 4*ob + reg: its 8 consecutive outputs
        f32x2 Z[S][4], P[S][4];                              // pair h of block ob: o = 8*kq + 4*ob + 2*h + {0,1}
        {
            const int oa0 = 8 * (r16 >> 2) + (r16 & 3);      // A-fragment row -> output column (block 0); +4 for block 1
            // The W fragments of support s + 1 are requested before the MFMAs of support s are issued, and the two output
            // blocks' chains are interleaved: as first written (read, wait, three dependent MFMAs, per block) every
            // step exposed a full LDS round trip and the MFMA latency (r02a: Z = 3.6 k cycles for 1.5 k of matrix work).
            bf16x8 wh[2][2], wl[2][2];                       // [stage][ob]
            auto frag = [&](int s, int st) {
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) {
                    const int oa = oa0 + 4 * ob;
                    const int off = (s * 32 + oa) * 32 + (((kq ^ gml_wkey(oa)) & 3) << 3);
                    wh[st][ob] = *reinterpret_cast<const bf16x8*>(Wof_h + off);
                    wl[st][ob] = *reinterpret_cast<const bf16x8*>(Wof_l + off);
                }
            };
            frag(0, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int st = s & 1;
                if (s + 1 < S) frag(s + 1, st ^ 1);
                f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = f32x4{0.f, 0.f, 0.f, 0.f};
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[st][0], xh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[st][1], xh, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[st][0], xl, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[st][1], xl, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[st][0], xh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[st][1], xh, d1, 0, 0, 0);
                Z[s][0] = f32x2{d0[0], d0[1]}; Z[s][1] = f32x2{d0[2], d0[3]};
                Z[s][2] = f32x2{d1[0], d1[1]}; Z[s][3] = f32x2{d1[2], d1[3]};
#pragma unroll
                for (int h = 0; h < 4; ++h) P[s][h] = f32x2{0.f, 0.f};
            }
            // pin the order the scheduler would otherwise undo: 4 reads, then per support (4 reads of the next, 6 MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
        }

        GML_T(2);
        // ---- edge phase (fp32 VALU, packed): P += val * G[dst],  d[s] = <Z[s], G[dst]>
        for (int k = kbeg; k < kend; ++k) {
            const int dstl = col_l[k];
            float ev[S];
            gml_load_row<S, VAL_ALIGN>(ea_l + k * S, ev);
            f32x2 gv[4];
            {
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(gs + dstl * LDG + 8 * kq);
                const f32x4 t1 = *reinterpret_cast<const f32x4*>(gs + dstl * LDG + 8 * kq + 4);
                gv[0] = f32x2{t0.x, t0.y}; gv[1] = f32x2{t0.z, t0.w};
                gv[2] = f32x2{t1.x, t1.y}; gv[3] = f32x2{t1.z, t1.w};
            }
            float d[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f32x2 a2 = f32x2{0.f, 0.f};
                const f32x2 e2 = f32x2{ev[s], ev[s]};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    P[s][h] = e2 * gv[h] + P[s][h];
                    a2 = Z[s][h] * gv[h] + a2;
                }
                d[s] = a2.x + a2.y;
            }
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
        GML_T(3);
        __syncthreads();                                     // dval rows complete; G window no longer needed
        GML_T(4);
        // The old dx values (accumulate mode) first, then the next group's loads: both BEFORE this group's dval / dX
        // stores (memory operations complete in order: behind the stores the loads would first wait for those to
        // drain), the latter in flight through the dX and dW phases.  All unconditional and clamped, so the dX
        // chain's wait for the old values is an exact count that leaves the prefetch in flight.
        latch();
        issue(min(g + 1, g1 - 1));
        GML_T(8);

        if (p.dval) {
            if constexpr (S % 4 == 0) {
                f32x4* dst = reinterpret_cast<f32x4*>(p.dval + (int64_t)kb * S);
                for (int i = tid; i < ne * (S / 4); i += 512) dst[i] = reinterpret_cast<const f32x4*>(ea_l)[i];
            } else {
                for (int i = tid; i < ne * S; i += 512) p.dval[(int64_t)kb * S + i] = ea_l[i];
            }
        }

        GML_T(9);
        // P -> bf16 (hi, lo) once; the pairs are the A fragments of dX and the inputs of the dW transposes
        bf16x8 PH[S], PL[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float pv[8] = {P[s][0].x, P[s][0].y, P[s][1].x, P[s][1].y, P[s][2].x, P[s][2].y, P[s][3].x, P[s][3].y};
            gml_split8(pv, PH[s], PL[s]);
        }
        GML_T(10);
        // ---- dX = P W^T: one K=32 step per support (k = o = 8*kq + i)
        if (p.dx) {
            if (!(p.flags & GML_ACCUM)) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // fragments of support s + 1 in flight while the MFMAs of support s run (see the Z projection)
            bf16x8 vh[2][NFB], vl[2][NFB];
            auto fragx = [&](int s, int st) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    const int ff = fb * 16 + r16;                              // B[k = o][j = f]: 8 consecutive o of row f
                    const int off = (s * 32 + ff) * 32 + (((kq ^ gml_wkey(ff)) & 3) << 3);
                    vh[st][fb] = *reinterpret_cast<const bf16x8*>(Wfo_h + off);
                    vl[st][fb] = *reinterpret_cast<const bf16x8*>(Wfo_l + off);
                }
            };
            fragx(0, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int st = s & 1;
                if (s + 1 < S) fragx(s + 1, st ^ 1);
                const bf16x8 ph = PH[s], pl = PL[s];
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vh[st][fb], dxa[fb], 0, 0, 0);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vl[st][fb], dxa[fb], 0, 0, 0);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vh[st][fb], dxa[fb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NFB, 1);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NFB, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NFB, 1);
            }
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) {
                const int f = fb * 16 + r16;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int lr = (int)((out_rows >> (8 * reg)) & 255u);
                    if (f < p.Fin && lr < nr) p.dx[(r0 + lr) * p.lddx + f] = dxa[fb][reg];
                }
            }
        }

        GML_T(5);
        // ---- dW += X^T P over the 128 rows of the group
        if (p.dw_partial) {
            // X^T / P^T tiles (bf16 hi, lo) [f or o][row] for the row contraction.  The 16 x 32 register tile of a wave
            // (lane (row, kq): 8 consecutive features) is transposed ON THE MATRIX CORES: tile x [0/1 selection of one
            // 16-wide block] leaves, in lane (column n, g), rows 4g..4g+3 of feature 16*blk + n -- exact (the values
            // are bf16 numbers) -- which go to LDS as ONE 8-byte store per block and image instead of eight 2-byte ones.
            bf16x8 sel[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                uint32_t u[4];
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const int f0 = 8 * kq + 2 * j2 - 16 * blk;
                    u[j2] = (f0 == r16 ? 0x3F80u : 0u) | (f0 + 1 == r16 ? 0x3F800000u : 0u);
                }
                sel[blk] = __builtin_bit_cast(bf16x8, u32x4_t{u[0], u[1], u[2], u[3]});
            }
            auto put_t = [&](const bf16x8 a, __bf16* dst_row0) {     // dst_row0: image + first channel row of block 0
                const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, sel[blk], z4, 0, 0, 0);
                    const bf16x2 p01 = __builtin_convertvector(f32x2{t[0], t[1]}, bf16x2);
                    const bf16x2 p23 = __builtin_convertvector(f32x2{t[2], t[3]}, bf16x2);
                    uint32_t w2[2] = {__builtin_bit_cast(uint32_t, p01), __builtin_bit_cast(uint32_t, p23)};
                    // rows wave*16 + 4*kq .. +3 of channel ch: K step wave >> 1, 16-byte chunk 2*(wave & 1) + (kq >> 1)
                    const int ch = 16 * blk + r16;
                    const int c16 = (2 * (wave & 1) + (kq >> 1)) ^ ((ch >> 2) & 3);
                    *reinterpret_cast<uint2*>(dst_row0 + ch * LDT + 32 * (wave >> 1) + 8 * c16 + 4 * (kq & 1)) = uint2{w2[0], w2[1]};
                }
            };
            put_t(xh, xT_h);
            put_t(xl, xT_l);
#pragma unroll
            for (int sl = 0; sl < C::NSLAB; ++sl) {
                __syncthreads();                             // slab buffer free: the dval copy-out (sl == 0) or the
                                                             // previous slab's fragment reads are done in every wave
#pragma unroll
                for (int se = 0; se < C::SE; ++se) {
                    const int s = sl * C::SE + se;
                    put_t(PH[s], pT_h + se * 32 * LDT);
                    put_t(PL[s], pT_l + se * 32 * LDT);
                }
                __syncthreads();
                if (wave < C::SE * NFB * 2) {
                    const int ob = wave & 1, fb = (wave >> 1) % NFB, se = (wave >> 1) / NFB;
                    // two independent accumulator chains (hi.lo + lo.hi, hi.hi) and the next K step's fragments in
                    // flight: the single chain of 12 dependent MFMAs behind 16 serial reads was latency, not work
                    f32x4 d = dwacc[sl], d2 = f32x4{0.f, 0.f, 0.f, 0.f};
                    bf16x8 fa[2][2], fbv[2][2];                // [stage][hi, lo]
                    auto fragw = [&](int st, int sg) {
                        const int kx = kq ^ (((fb * 16 + r16) >> 2) & 3), kp = kq ^ (((ob * 16 + r16) >> 2) & 3);
                        const int xo = (fb * 16 + r16) * LDT + 32 * st + 8 * kx;                 // A[i = f][k = row]
                        const int po = (se * 32 + ob * 16 + r16) * LDT + 32 * st + 8 * kp;       // B[k = row][j = o]
                        fa[sg][0] = *reinterpret_cast<const bf16x8*>(xT_h + xo);
                        fa[sg][1] = *reinterpret_cast<const bf16x8*>(xT_l + xo);
                        fbv[sg][0] = *reinterpret_cast<const bf16x8*>(pT_h + po);
                        fbv[sg][1] = *reinterpret_cast<const bf16x8*>(pT_l + po);
                    };
                    fragw(0, 0);
#pragma unroll
                    for (int st = 0; st < ROWS / 32; ++st) {
                        const int sg = st & 1;
                        if (st + 1 < ROWS / 32) fragw(st + 1, sg ^ 1);
                        d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[sg][1], fbv[sg][0], d2, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[sg][0], fbv[sg][0], d, 0, 0, 0);
                        d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[sg][0], fbv[sg][1], d2, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 2);
#pragma unroll
                    for (int st = 0; st < ROWS / 32; ++st) {
                        if (st + 1 < ROWS / 32) __builtin_amdgcn_sched_group_barrier(0x100, 4, 2);
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 2);
                    }
                    d += d2;
                    dwacc[sl] = d;
                }
            }
        }
    }

    GML_T(6);
    // ---- one dW partial per workgroup: wave w holds, per slab, block (se, fb, ob): D[i = f][j = o]
    if (p.dw_partial && g0 < g1 && wave < C::SE * NFB * 2) {
        float* out = p.dw_partial + (int64_t)wg * S * p.Fin * p.Fout;
        const int ob = wave & 1, fb = (wave >> 1) % NFB, se = (wave >> 1) / NFB;
        const int o = ob * 16 + r16;
#pragma unroll
        for (int sl = 0; sl < C::NSLAB; ++sl) {
            const int s = sl * C::SE + se;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int f = fb * 16 + 4 * kq + reg;
                if (f < p.Fin && o < p.Fout) out[((int64_t)s * p.Fin + f) * p.Fout + o] = dwacc[sl][reg];
            }
        }
    }
}

template <int S, int NFB>
int gml_launch_bwd2(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st);

#define GML_DEFINE_BWD2(SV, NFBV)                                                                            \
    template <>                                                                                              \
    int gml_launch_bwd2<SV, NFBV>(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st) {            \
        static_assert(GmlBwd2Cfg<SV, NFBV>::OK, "S must be even");                                           \
        GML_ALLOW_BIG_LDS(rc1, (&gml_k_spectconv_bwd2<SV, NFBV, true>), 160 * 1024) \
        GML_ALLOW_BIG_LDS(rc0, (&gml_k_spectconv_bwd2<SV, NFBV, false>), 160 * 1024) \
        if (rc1 != hipSuccess) return (int)rc1;                                                              \
        if (rc0 != hipSuccess) return (int)rc0;                                                              \
        if (p.xvec) hipLaunchKernelGGL((gml_k_spectconv_bwd2<SV, NFBV, true>), grid, dim3(512), lds, st, p); \
        else hipLaunchKernelGGL((gml_k_spectconv_bwd2<SV, NFBV, false>), grid, dim3(512), lds, st, p);       \
        return gml_launch_status();                                                                          \
    }
