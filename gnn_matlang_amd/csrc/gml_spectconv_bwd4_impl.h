// Fused backward, bf16x3, third layout with an LDS-DMA landing ring (S in {4, 8}, Fin <= 32, 16 < Fout <= 32, float4-addressable
// x / g rows, every group inside the staging capacities): same outputs as gml_k_spectconv_bwd3 (gml_spectconv_bwd3_impl.h)
//
//   dX = sum_s A_s (G W_s^T),   dval[e,s] = < X[src] W_s, G[dst] >,   dW_s = X^T (A_s G)      (autograd of libs/spect_conv.py:76-80)
//
// What changed against bwd3 (profiles/r02_f_bwd3_phases.txt: issue of the next group's 13 register loads
// 12.5 %, their commit to LDS 10 %, the barriers around the commit 10 % of the wave time):
//   * a group's row pointers, column ids, value rows, G window, dz rows and record land in the OTHER of two LDS slots by
//     `buffer_load_dwordx4 ... lds` (1 KiB per wave-instruction, no VGPR, no ds_write); the waves deal the ~60 instructions
//     of a group among themselves right after the edge phase, and the group's top is one `s_waitcnt vmcnt(0)` + `s_barrier`
//     instead of barrier + commit + barrier.  The ~35 prefetch registers are gone;
//   * with them gone the edge loop is software pipelined (operands of edge k + 1 and the column id of edge k + 2 requested
//     before the arithmetic of edge k, two register sets, no rotation moves) -- the form that spilled in bwd3;
//   * layouts as in the forward (gml_spectconv_fwd3_impl.h): verbatim copies from the 16-byte aligned edge kb & ~3, absolute
//     column ids, G window in 8-row blocks 1040 bytes apart from row lo & ~7; out-of-range protection by buffer descriptors;
//   * the dW phase's bf16 images (X: 16 KB, P: two supports per slab = 32 KB) live in the CURRENT slot once its value rows
//     have been copied out as dval: W image + two slots is all the LDS the kernel has.
#pragma once
#include "gml_common.h"
#include "gml_spectconv_bwd3_impl.h"
#include "gml_spectconv_fwd3_impl.h"      // gml_dma16, gml_raw_rsrc

template <int S, int NFB>
struct GmlBwd4Cfg {
    static constexpr int ROWS = 128, NT = 512, NW = 8;
    static constexpr int XCAP = 200;                           // staged G-window rows incl. the <= 7 rows of alignment slack
    static constexpr int XBLK = 1040;
    static constexpr int W_HALF = S * 32 * 32;
    static constexpr int W_BYTES = 4 * W_HALF;
    static constexpr int WM_BYTES = 512;                       // DZ: [4][32] rows of wmix
    static constexpr int REC_BYTES = 4 * 256;
    static constexpr int RP_BYTES = 528, DZ_BYTES = 2048;
    static constexpr int G_BYTES = XCAP / 8 * XBLK;
    static constexpr int AVAIL = 160 * 1024 - W_BYTES - WM_BYTES - REC_BYTES - 2 * (RP_BYTES + DZ_BYTES + G_BYTES);
    static constexpr int ECAP_RAW = AVAIL / (2 * (4 + 4 * S));
    static constexpr int ECAP = ECAP_RAW >= 1024 ? 1024 : ECAP_RAW / 64 * 64;      // S = 8: 960
    static constexpr int COL_BYTES = ECAP * 4, VAL_BYTES = ECAP * S * 4;
    // dW phase images inside the current slot
    static constexpr int XT_BYTES = 2 * ROWS * 64;             // X hi, lo   [position][32 f]
    static constexpr int SS = 2;                               // supports per dW slab
    static constexpr int NSLAB = S / SS;
    static constexpr int PT_BYTES = 2 * SS * ROWS * 64;        // P hi, lo   [se][position][32 o]
    static constexpr int NBLK = SS * NFB * 2;                  // 16 x 16 output blocks of a slab: (se, fb, ob)
    static constexpr int DATA_BYTES = RP_BYTES + DZ_BYTES + COL_BYTES + VAL_BYTES + G_BYTES;
    static constexpr int SLOT_BYTES = DATA_BYTES > XT_BYTES + PT_BYTES ? DATA_BYTES : XT_BYTES + PT_BYTES;
    static constexpr int OFF_WM = W_BYTES, OFF_REC = OFF_WM + WM_BYTES, OFF_SLOT = OFF_REC + REC_BYTES;
    static constexpr int OFF_DZ = RP_BYTES, OFF_COL = OFF_DZ + DZ_BYTES, OFF_VAL = OFF_COL + COL_BYTES, OFF_G = OFF_VAL + VAL_BYTES;
    static constexpr size_t lds_bytes() { return (size_t)OFF_SLOT + 2 * (size_t)SLOT_BYTES; }
    static constexpr bool OK = (S % SS == 0) && (NFB == 1 || NFB == 2) && NBLK <= NW && lds_bytes() <= 160 * 1024;
    static_assert(SLOT_BYTES % 16 == 0 && OFF_SLOT % 16 == 0, "16-byte aligned landing zones");
};

// DZ: dx starts from dz[row] . wmix (see GmlBwdParams) instead of zero
template <int S, int NFB, bool DZ>
__global__ __launch_bounds__(512, 1) void gml_k_spectconv_bwd4(const GmlBwdParams p) {
    using C = GmlBwd4Cfg<S, NFB>;
    static_assert(C::OK && S % 4 == 0, "unsupported shape");
    constexpr int ROWS = C::ROWS, NT = C::NT, NW = C::NW, SS = C::SS;
    constexpr int VROW = S * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* W_h = reinterpret_cast<__bf16*>(lds_raw);        // [s][o][f], chunks XOR gml_wkey3(o)
    __bf16* W_l = W_h + C::W_HALF;
    float* wm_l = reinterpret_cast<float*>(lds_raw + C::OFF_WM);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);
    if (g0 >= g1) return;

    if constexpr (DZ) {
        if (tid < 128) wm_l[tid] = ((tid >> 5) < p.nmix && (tid & 31) < p.Fin) ? ((tid >> 5) < p.nmix1 ? p.wmix[(tid >> 5) * p.Fin + (tid & 31)] : p.wmix2[((tid >> 5) - p.nmix1) * p.Fin + (tid & 31)]) : 0.f;
    }
    // W -> bf16 (hi, lo) image, zero padded to 32 x 32
    for (int e = tid; e < S * 32 * 32; e += NT) {
        const int f = e & 31, o = (e >> 5) & 31, s = e >> 10;
        const float v = (f < p.Fin && o < p.Fout) ? p.w[((int64_t)s * p.Fin + f) * p.Fout + o] : 0.f;
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        const int i = (s * 32 + o) * 32 + ((((f >> 3) ^ gml_wkey3(o)) & 3) << 3) + (f & 7);
        W_h[i] = h; W_l[i] = l;
    }
    const int etot = p.rowptr[p.nrows];

    f32x4 dwacc[C::NSLAB];
#pragma unroll
    for (int i = 0; i < C::NSLAB; ++i) dwacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- buffer descriptors (hardware range check: out-of-range lanes read zeros) and the landing-zone addresses
    const uint32_t lds0 = (uint32_t)(uintptr_t)((gml_lds_void*)lds_raw);
    const u32x4 rs_rec = gml_raw_rsrc(p.ginfo, (uint32_t)p.ngroups * (GML_GREC_INTS(128) * 4));
    const u32x4 rs_rp = gml_raw_rsrc(p.rowptr, (uint32_t)(p.nrows + 1) * 4u);
    const u32x4 rs_g = gml_raw_rsrc(p.g, (uint32_t)(p.nrows * p.ldg) * 4u);
    const u32x4 rs_dz = gml_raw_rsrc(p.dz, DZ ? (uint32_t)p.nrows * 16u : 0u);
    const int ldgb = (int)p.ldg * 4;

    struct Geo { int kb, kb4, ne, ne4, lo8, nwin8; };
    auto geo_of = [&](int g) -> Geo {
        const int4 v = *reinterpret_cast<const int4*>(lds_raw + C::OFF_REC + (g & 3) * 256);
        Geo q;
        q.kb = __builtin_amdgcn_readfirstlane(v.x); q.ne = __builtin_amdgcn_readfirstlane(v.y);
        const int lo = __builtin_amdgcn_readfirstlane(v.z), nwin = __builtin_amdgcn_readfirstlane(v.w);
        q.kb4 = q.kb & ~3; q.ne4 = q.ne + (q.kb & 3);
        q.lo8 = lo & ~7; q.nwin8 = q.ne > 0 ? lo + nwin - q.lo8 : 0;
        return q;
    };
    auto dma_rec = [&](int g) {
        if (wave == 7 && lane < 9) gml_dma16(rs_rec, lds0 + C::OFF_REC + (g & 3) * 256, g * (GML_GREC_INTS(128) * 4) + lane * 16);
    };
    float xb[8];                                             // own x row of the NEXT group (plain loads, in flight over the dW phase)
    // The DMA instructions of group g (-> slot g & 1, record of g + 2) are issued in three parts spread over the second half
    // of the previous trip (value rows | G window | the small arrays): every wave issuing its whole share at one point
    // saturates the CU's address unit and the waves stall in the issue (measured: 10 % of the wave time).
    auto issue_a = [&](int g, const Geo& q) {                // own x row (plain loads) + value rows
        const uint32_t slot = lds0 + C::OFF_SLOT + (g & 1) * C::SLOT_BYTES;
        const int64_t r0 = (int64_t)g * ROWS;
        {
            const unsigned char* rec = lds_raw + C::OFF_REC + (g & 3) * 256;
            const int row_n = rec[16 + wave * 16 + r16];
            const float* xr = p.x + min(r0 + row_n, p.nrows - 1) * p.ldx;
            const int f4max = (p.Fin + 3) / 4 * 4 - 4;
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(xr + min(8 * kq + 4 * q4, f4max));
                xb[4 * q4] = t.x; xb[4 * q4 + 1] = t.y; xb[4 * q4 + 2] = t.z; xb[4 * q4 + 3] = t.w;
            }
        }
        if (!(GML_ABL & 64)) {
            // value rows through a descriptor based at the group's first staged edge: no 4 GB limit on the array
            const uint64_t left = (uint64_t)(etot - q.kb4);
            const u32x4 rs_val = gml_raw_rsrc(p.val + (int64_t)q.kb4 * S, (uint32_t)min(left * VROW, (uint64_t)0xffffff00u));
            constexpr int EPI = 1024 / VROW;                 // value rows per instruction
            const int nvi = (q.ne4 + EPI - 1) / EPI;
            for (int j = wave; j < nvi; j += NW)
                if (j * 1024 + lane * 16 < q.ne4 * VROW) gml_dma16(rs_val, slot + C::OFF_VAL + j * 1024, j * 1024 + lane * 16);
        }
        asm volatile("" ::: "memory");
    };
    auto issue_b = [&](int g, const Geo& q) {                // G window
        const uint32_t slot = lds0 + C::OFF_SLOT + (g & 1) * C::SLOT_BYTES;
        if (!(GML_ABL & 64)) {
            const int nxi = (q.nwin8 + 7) >> 3;
            const int c4 = (lane & 7) * 4;
            for (int i = (wave + 3) & 7; i < nxi; i += NW) {
                const int rr = q.lo8 + 8 * i + (lane >> 3);
                gml_dma16(rs_g, slot + C::OFF_G + i * C::XBLK, rr * ldgb + c4 * 4);   // (all 8 chunks: see the prologue note)
            }
        }
        asm volatile("" ::: "memory");
    };
    auto issue_c = [&](int g, const Geo& q) {                // column ids, row pointers, dz rows, record of g + 2
        const uint32_t slot = lds0 + C::OFF_SLOT + (g & 1) * C::SLOT_BYTES;
        const int r0 = g * ROWS;
        if (!(GML_ABL & 64)) {
            const uint64_t left = (uint64_t)(etot - q.kb4);
            const u32x4 rs_col = gml_raw_rsrc(p.col + q.kb4, (uint32_t)min(left * 4u, (uint64_t)0xffffff00u));
            const int nci = (q.ne4 + 255) >> 8;
            if (wave < nci && 256 * wave + 4 * lane < q.ne4) gml_dma16(rs_col, slot + C::OFF_COL + wave * 1024, (256 * wave + 4 * lane) * 4);
        }
        if (wave == 4 && lane < 33) gml_dma16(rs_rp, slot, (r0 + 4 * lane) * 4);
        if constexpr (DZ) {
            if (wave == 5 || wave == 6) gml_dma16(rs_dz, slot + C::OFF_DZ + (wave - 5) * 1024, (r0 + (wave - 5) * 64 + lane) * 16);
        }
        if (g + 2 < g1) dma_rec(g + 2);
        asm volatile("" ::: "memory");
    };
    auto issue = [&](int g, const Geo& q) { issue_a(g, q); issue_b(g, q); issue_c(g, q); };

    // ---- prologue: records g0 / g0 + 1, data of g0.  (The G window is always copied 32 floats wide: columns at or beyond
    //      Fout are g's zero padding, or -- when ldg < 32 -- finite values of the next row; they only ever meet the zero rows
    //      of the W image.  Nothing relies on LDS contents a DMA did not write: the dW images reuse the slots.)
    dma_rec(g0);
    if (g0 + 1 < g1) dma_rec(g0 + 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(g0, geo_of(g0));

#ifdef GML_BWD2_TIMING
    unsigned tacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tprev_ = (unsigned)__builtin_readcyclecounter();
#endif
    for (int g = g0; g < g1; ++g) {
        // the group's data (this wave's share) has landed once nothing is outstanding -- the only younger operations are the
        // previous group's dval / dx stores, issued a whole dW phase ago; the barrier extends that to the other waves'
        // shares and says that every wave has left the previous group's dW phase (the slot the next DMAs overwrite)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        GML_T3(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        GML_T3(1);
        const Geo q = geo_of(g);
        unsigned char* slot = lds_raw + C::OFF_SLOT + (g & 1) * C::SLOT_BYTES;
        const int* rp_l = reinterpret_cast<const int*>(slot);
        const int* col_l = reinterpret_cast<const int*>(slot + C::OFF_COL);
        float* ea_l = reinterpret_cast<float*>(slot + C::OFF_VAL);
        const unsigned char* rec = lds_raw + C::OFF_REC + (g & 3) * 256;
        const int row = rec[16 + wave * 16 + r16];
        const int64_t r0 = (int64_t)g * ROWS;
        const int nr = (int)min((int64_t)ROWS, p.nrows - r0);
        const bool rvalid = row < nr;
        const int xoff = C::OFF_SLOT + (g & 1) * C::SLOT_BYTES + C::OFF_G + kq * 32 - q.lo8 * 130;

        f32x4 dxa[NFB];
        f32x4 dzv = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (DZ) dzv = *reinterpret_cast<const f32x4*>(slot + C::OFF_DZ + row * 16);
        const int kbeg = rvalid ? rp_l[row] - q.kb4 : 0;
        const int kend = rvalid ? rp_l[row + 1] - q.kb4 : 0;

        bf16x8 xh, xl;                                       // own X row, features 8*kq .. 8*kq+7: B fragment of Z^T, row of the X image
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (!(rvalid && 8 * kq + j < p.Fin)) xb[j] = 0.f;
        gml_split8(xb, xh, xl);

        // ---- Z^T = W^T X^T: MFMA row i of block ob carries o = 8*(i>>2) + 4*ob + (i&3), so lane kq receives its 8
        //      consecutive outputs o = 8*kq + 4*ob + reg.  Fragments of support s + 1 are requested before the MFMAs of s.
        f32x2 Z[S][4], P[S][4];
        {
            const int oa0 = 8 * (r16 >> 2) + (r16 & 3);
            bf16x8 wh[2][2], wl[2][2];
            auto frag = [&](int s, int st) {
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) {
                    const int oa = oa0 + 4 * ob;
                    const int off = (s * 32 + oa) * 32 + (((kq ^ gml_wkey3(oa)) & 3) << 3);
                    wh[st][ob] = *reinterpret_cast<const bf16x8*>(W_h + off);
                    wl[st][ob] = *reinterpret_cast<const bf16x8*>(W_l + off);
                }
            };
            if (!(GML_ABL & 16)) frag(0, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int st = s & 1;
                if (GML_ABL & 16) {
#pragma unroll
                    for (int h = 0; h < 4; ++h) { Z[s][h] = f32x2{xb[h], xb[h + 4]}; P[s][h] = f32x2{0.f, 0.f}; }
                    continue;
                }
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
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
        }
        GML_T3(2);

        // ---- edge phase (fp32 VALU, packed): P += val * G[dst],  d[s] = <Z[s], G[dst]>, software pipelined: the operands of
        //      edge k + 1 and the column id of edge k + 2 are requested before the arithmetic of edge k (two register sets)
        {
            struct Ops { f32x4 e[S / 4]; f32x4 t0, t1; };
            auto fetch = [&](Ops& o, int kk, int c) {
#pragma unroll
                for (int i = 0; i < S / 4; ++i) o.e[i] = *reinterpret_cast<const f32x4*>(ea_l + kk * S + 4 * i);
                const int off = xoff + c * 128 + ((c & ~7) << 1);
                o.t0 = *reinterpret_cast<const f32x4*>(lds_raw + off);
                o.t1 = *reinterpret_cast<const f32x4*>(lds_raw + off + 16);
            };
            auto edge = [&](const Ops& o, int kk) {
                const f32x2 gv[4] = {f32x2{o.t0.x, o.t0.y}, f32x2{o.t0.z, o.t0.w}, f32x2{o.t1.x, o.t1.y}, f32x2{o.t1.z, o.t1.w}};
                float d[S];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    f32x2 a2 = f32x2{0.f, 0.f};
                    const float ev = o.e[s >> 2][s & 3];
                    const f32x2 e2 = f32x2{ev, ev};
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        P[s][h] = e2 * gv[h] + P[s][h];
                        a2 = Z[s][h] * gv[h] + a2;
                    }
                    d[s] = a2.x + a2.y;
                }
#pragma unroll
                for (int c = 0; c < S / 4; ++c) {
                    const auto a01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[4 * c]), __float_as_uint(d[4 * c + 1]), false, false);
                    const auto a23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[4 * c + 2]), __float_as_uint(d[4 * c + 3]), false, false);
                    const float c01 = __uint_as_float(a01[0]) + __uint_as_float(a01[1]);
                    const float c23 = __uint_as_float(a23[0]) + __uint_as_float(a23[1]);
                    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(c01), __float_as_uint(c23), false, false);
                    ea_l[kk * S + 4 * c + kq] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
                }
            };
            int k = kbeg;
            const int ke = (GML_ABL & 4) ? kbeg : kend;
            if (k < ke) {
                Ops A, B;
                const int klast = ke - 1;
                fetch(A, k, col_l[k]);
                int cn = col_l[min(k + 1, klast)];
                for (;;) {
                    const int c2 = col_l[min(k + 2, klast)];
                    fetch(B, min(k + 1, klast), cn);
                    edge(A, k);
                    if (++k >= ke) break;
                    cn = col_l[min(k + 2, klast)];
                    fetch(A, min(k + 1, klast), c2);
                    edge(B, k);
                    if (++k >= ke) break;
                }
            }
        }
        GML_T3(3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // dval rows complete; G window no longer needed
        asm volatile("" ::: "memory");
        GML_T3(4);
        // (lane-only address terms of the phases below: recomputed per group from an opaque copy of the thread id, see bwd3)
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const int r16o = tid_o & 15, kqo = (tid_o >> 4) & 3;
        // P -> bf16 (hi, lo) once: B fragments of dX^T (k = o = 8*kq + j) and the rows of the P image
        bf16x8 PH[S], PL[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float pv[8] = {P[s][0].x, P[s][0].y, P[s][1].x, P[s][1].y, P[s][2].x, P[s][2].y, P[s][3].x, P[s][3].y};
            gml_split8(pv, PH[s], PL[s]);
        }
        GML_T3(10);
        const int gn = (GML_ABL & 1) ? g0 : g + 1;
        Geo qn = q;
        if (g + 1 < g1) { qn = geo_of(gn); issue_a(gn, qn); }
        GML_T3(8);

        if (p.dval && !(GML_ABL & 8)) {                      // dval rows of this group's edges (the slot starts kb - kb4 rows early)
            f32x4* dst = reinterpret_cast<f32x4*>(p.dval + (int64_t)q.kb * S);
            const f32x4* srcv = reinterpret_cast<const f32x4*>(ea_l + (q.kb - q.kb4) * S);
            for (int i = tid_o; i < q.ne * (S / 4); i += NT) dst[i] = srcv[i];
        }
        if (g + 1 < g1) issue_b(gn, qn);
        GML_T3(9);

        // ---- dX^T = W P^T: A[i = f][k = o] = W_s[f][o] comes transposed out of the [s][o][f] image (see bwd3)
        if (p.dx && !(GML_ABL & 32)) {
            if constexpr (DZ) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) a += dzv[q4] * *reinterpret_cast<const f32x4*>(wm_l + q4 * 32 + 16 * fb + 4 * kqo);
                    dxa[fb] = a;
                }
            } else {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const int tj = r16o >> 2, tc = r16o & 3;
            int aoff[2][NFB];                                // byte offsets inside one support's image
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    const int o = 8 * kqo + 4 * h + tj, cidx = 4 * fb + tc;
                    aoff[h][fb] = o * 64 + ((((cidx >> 1) ^ gml_wkey3(o)) & 3) << 4) + ((cidx & 1) << 3);
                }
            const unsigned char* Wh8 = reinterpret_cast<const unsigned char*>(W_h);
            const unsigned char* Wl8 = reinterpret_cast<const unsigned char*>(W_l);
            bf16x8 vh[2][NFB], vl[2][NFB];
            auto fragx = [&](int s, int st) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    vh[st][fb] = gml_tr_frag(Wh8 + s * 2048 + aoff[0][fb], Wh8 + s * 2048 + aoff[1][fb]);
                    vl[st][fb] = gml_tr_frag(Wl8 + s * 2048 + aoff[0][fb], Wl8 + s * 2048 + aoff[1][fb]);
                }
            };
            fragx(0, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int st = s & 1;
                if (s + 1 < S) fragx(s + 1, st ^ 1);
                const bf16x8 ph = PH[s], pl = PL[s];
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl[st][fb], ph, dxa[fb], 0, 0, 0);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[st][fb], pl, dxa[fb], 0, 0, 0);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[st][fb], ph, dxa[fb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * NFB, 1);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) __builtin_amdgcn_sched_group_barrier(0x100, 4 * NFB, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NFB, 1);
            }
            float* dr = p.dx + (r0 + row) * p.lddx + 4 * kqo;
            if (GML_ABL & 8) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) asm volatile("" :: "v"(dxa[fb]));
            } else if (p.dxvec) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
                    if (rvalid && 16 * fb + 4 * kqo < p.Fin) *reinterpret_cast<f32x4*>(dr + 16 * fb) = dxa[fb];
            } else {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        if (rvalid && 16 * fb + 4 * kqo + reg < p.Fin) dr[16 * fb + reg] = dxa[fb][reg];
            }
        }
        if (g + 1 < g1) issue_c(gn, qn);
        GML_T3(5);

        // ---- dW += X^T P over the rows of the group.  Row-major bf16 images [position = wave*16 + r16][32 channels] in the
        //      current slot (a lane's 8 channels = one 16-byte chunk, XOR gml_tkey3(position)); the contraction reads them
        //      transposed.  Two supports per slab, one 16 x 16 block (se, fb, ob) per wave, its K = 128 contraction split
        //      into two interleaved accumulator chains.
        if (p.dw_partial && !(GML_ABL & 2)) {
            unsigned char* xT = slot;                        // [hi, lo][position]     64-byte rows
            unsigned char* pT = slot + C::XT_BYTES;          // [hi, lo][se][position] 64-byte rows
            const int pos = wave * 16 + r16o;
            const int woff = pos * 64 + (((kqo ^ gml_tkey3(pos)) & 3) << 4);
            const int tj = r16o >> 2, tc = r16o & 3;
            int roff[2][2];                                  // [h][16-wide channel block]
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const int ps = 8 * kqo + 4 * h + tj, cidx = 4 * blk + tc;
                    roff[h][blk] = ps * 64 + ((((cidx >> 1) ^ gml_tkey3(ps)) & 3) << 4) + ((cidx & 1) << 3);
                }
            const int ob = wave & 1, fb = (wave >> 1) % NFB, se_w = (wave >> 1) / NFB;
            constexpr int KS = ROWS / 32;                    // K steps of the row contraction
#pragma unroll
            for (int sl = 0; sl < C::NSLAB; ++sl) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                // slab region free: the dval copy-out (sl == 0) or the previous
                asm volatile("" ::: "memory");               // slab's fragment reads are done in every wave
                if (sl == 0) {
                    *reinterpret_cast<bf16x8*>(xT + woff) = xh;
                    *reinterpret_cast<bf16x8*>(xT + ROWS * 64 + woff) = xl;
                }
#pragma unroll
                for (int se = 0; se < SS; ++se) {
                    const int s = sl * SS + se;
                    *reinterpret_cast<bf16x8*>(pT + se * ROWS * 64 + woff) = PH[s];
                    *reinterpret_cast<bf16x8*>(pT + (SS + se) * ROWS * 64 + woff) = PL[s];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                GML_T3(11);
                if (wave < C::NBLK) {
                    // every fragment of the slab is requested before the first MFMA (one LDS round trip per slab, not one
                    // per K step); one accumulator chain per K step, summed at the end
                    bf16x8 fah[KS], fal[KS], fbh[KS], fbl[KS];
#pragma unroll
                    for (int st = 0; st < KS; ++st) {
                        const unsigned char* xa = xT + st * 2048;
                        fah[st] = gml_tr_frag(xa + roff[0][fb], xa + roff[1][fb]);
                        fal[st] = gml_tr_frag(xa + ROWS * 64 + roff[0][fb], xa + ROWS * 64 + roff[1][fb]);
                        const unsigned char* pa = pT + se_w * ROWS * 64 + st * 2048;
                        fbh[st] = gml_tr_frag(pa + roff[0][ob], pa + roff[1][ob]);
                        fbl[st] = gml_tr_frag(pa + SS * ROWS * 64 + roff[0][ob], pa + SS * ROWS * 64 + roff[1][ob]);
                    }
                    f32x4 d[KS];
#pragma unroll
                    for (int st = 0; st < KS; ++st) d[st] = st == 0 ? dwacc[sl] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int st = 0; st < KS; ++st) d[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[st], fbh[st], d[st], 0, 0, 0);
#pragma unroll
                    for (int st = 0; st < KS; ++st) d[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[st], fbl[st], d[st], 0, 0, 0);
#pragma unroll
                    for (int st = 0; st < KS; ++st) d[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[st], fbh[st], d[st], 0, 0, 0);
                    dwacc[sl] = (d[0] + d[1]) + (d[2] + d[3]);
                }
                GML_T3(7);
            }
        }
    }

    GML_T3(6);
#ifdef GML_BWD2_TIMING
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) atomicAdd(&p.prof[i], (unsigned long long)tacc_[i]);
    }
#endif
    // ---- one dW partial per workgroup: block (se, fb, ob) of slab sl: D[i = f][j = o], lane (o = r16, kq): f = 16 fb + 4 kq + reg
    if (p.dw_partial && wave < C::NBLK) {
        float* out = p.dw_partial + (int64_t)wg * S * p.Fin * p.Fout;
        const int ob = wave & 1, fb = (wave >> 1) % NFB, se_w = (wave >> 1) / NFB;
        const int o = ob * 16 + r16;
#pragma unroll
        for (int sl = 0; sl < C::NSLAB; ++sl) {
            const int s = sl * SS + se_w;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int f = fb * 16 + 4 * kq + reg;
                if (f < p.Fin && o < p.Fout) out[((int64_t)s * p.Fin + f) * p.Fout + o] = dwacc[sl][reg];
            }
        }
    }
}

template <int S, int NFB>
int gml_launch_bwd4(const GmlBwdParams& p, dim3 grid, hipStream_t st);

#define GML_DEFINE_BWD4(SV, NFBV)                                                                            \
    template <>                                                                                              \
    int gml_launch_bwd4<SV, NFBV>(const GmlBwdParams& p, dim3 grid, hipStream_t st) {                        \
        static_assert(GmlBwd4Cfg<SV, NFBV>::OK, "unsupported shape");                                        \
        GML_ALLOW_BIG_LDS(rc1, (&gml_k_spectconv_bwd4<SV, NFBV, true>), 160 * 1024)                          \
        GML_ALLOW_BIG_LDS(rc0, (&gml_k_spectconv_bwd4<SV, NFBV, false>), 160 * 1024)                         \
        if (rc1 != hipSuccess) return (int)rc1;                                                              \
        if (rc0 != hipSuccess) return (int)rc0;                                                              \
        if (p.dz != nullptr) hipLaunchKernelGGL((gml_k_spectconv_bwd4<SV, NFBV, true>), grid, dim3(512), (GmlBwd4Cfg<SV, NFBV>::lds_bytes()), st, p);   \
        else hipLaunchKernelGGL((gml_k_spectconv_bwd4<SV, NFBV, false>), grid, dim3(512), (GmlBwd4Cfg<SV, NFBV>::lds_bytes()), st, p);                  \
        return gml_launch_status();                                                                          \
    }
