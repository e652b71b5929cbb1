// Stand-alone multi-support SpMM for ANY number of supports and row degrees up to ~16 per row on average:
//
//   H[r, s, :] = sum_{k in row r} val[k, s] x[col[k], :]       H [N, S, F] contiguous      (libs/spect_conv.py:77, the S propagate() calls)
//
// the kernel behind gml_spmm_fwd wherever 128-row group records exist (BASELINE.json's "SpMM HBM GB/s vs peak"; config 5 of
// the sr25 graphs -- 13 mask entries per row -- with S in {6, 12, 24, 48}).  Built like the forward's ring kernel
// (gml_spectconv_fwd3_impl.h): 8 compute waves + 4 loader waves, every byte enters LDS by `buffer_load ... lds`, one barrier
// per work item.  A work item = a block of 128 >> sh rows of a 128-row group with ALL S supports of its edges: the value rows
// are copied whole and contiguous ([edge][S], 1 KiB per DMA instruction -- a first version that gathered 16-byte chunks out
// of the rows re-fetched every 128-byte line S / 4 times: 0.28 of the roof at S = 48) into one of two 44 KB buffers, sh chosen
// per group so that a block's rows fit; the group's row pointers, column ids and X window land once per group.  Inside an
// item the waves deal themselves the work units (8-row tile, chunk of 8 supports): accumulators stay at 8 x 4 per lane
// whatever S is.  The 8 lanes of a row are adjacent and keep 4 features each: every 16-byte store instruction writes whole
// 128-byte (row, support) lines, no transposition needed; non-temporal (H is written once, far larger than the caches).
#pragma once
#include "gml_common.h"
#include "gml_spectconv_fwd3_impl.h"      // gml_dma16, gml_raw_rsrc

// 4 bytes per active lane: rs[voff] -> LDS lds_addr + 4 * lane
__device__ __forceinline__ void gml_dma4(u32x4 rs, uint32_t lds_addr, int voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}

struct GmlSpmm3Params {
    const int32_t* rowptr;
    const int32_t* col;
    const int32_t* ginfo;       // 128-row group records
    const float* val;           // [E, S] rows
    const float* x;             // rows float4-addressable
    int64_t ldx;
    float* h;                   // [N, S, hs]: this launch writes features hf0 .. hf0 + Fin - 1 of every support
    int64_t nrows;
    int32_t S;
    int32_t Fin, hs, hf0;
    int32_t ngroups, groups_per_wg;
};

// W48: feature chunks of up to 48 (x rows of 192 bytes, 16 rows per 3,088-byte window block as in gml_k_spectconv_fwd4's FB = 1 form):
// a 48-feature input is ONE pass over the edges instead of a 32- and a 16-feature launch that each read every value row (round 4).
template <bool W48>
struct GmlSpmm3CfgT {
    static constexpr int ROWS = 128, NLOAD = 4, NT = 512 + 64 * NLOAD;
    static constexpr int XRB = W48 ? 16 : 8;                    // rows per window block
    static constexpr int XROW = W48 ? 192 : 128;                // bytes of a staged x row
    static constexpr int XBLK = XRB * XROW + 16;                // 1040 / 3088: the 16-byte pad spreads the gathers over the bank groups
    static constexpr int XCAP = W48 ? 208 : 200;
    static constexpr int ECAP = 2048;                           // staged column ids per group (16 per row on average)
    static constexpr int VAL_BYTES = (W48 ? 30 : 44) * 1024;    // one value buffer: whole [edge][S] rows of a row block
    static constexpr int REC_BYTES = 4 * 256, RP_BYTES = 528;
    static constexpr int COL_BYTES = ECAP * 4, X_BYTES = XCAP / XRB * XBLK;
    static constexpr int SLOT_BYTES = RP_BYTES + COL_BYTES + X_BYTES;
    static constexpr int OFF_REC = 0, OFF_VAL = REC_BYTES, OFF_SLOT = OFF_VAL + 2 * VAL_BYTES;
    static constexpr int OFF_COL = RP_BYTES, OFF_X = OFF_COL + COL_BYTES;
    static constexpr size_t lds_bytes() { return (size_t)OFF_SLOT + 2 * (size_t)SLOT_BYTES; }
    static_assert(OFF_SLOT + 2 * SLOT_BYTES <= 160 * 1024 && SLOT_BYTES % 16 == 0, "LDS budget");
};
typedef GmlSpmm3CfgT<false> GmlSpmm3Cfg;

// VA: alignment class of the value rows in floats (4: S % 4 == 0 -> one 16-byte read per chunk; 2; 1)
template <int VA, bool W48 = false>
__global__ __launch_bounds__(GmlSpmm3Cfg::NT, 1) void gml_k_spmm3(const GmlSpmm3Params p) {
    using C = GmlSpmm3CfgT<W48>;
    constexpr int ROWS = C::ROWS, ECAP = C::ECAP, XCAP = C::XCAP, NL = C::NLOAD;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);
    if (g0 >= g1) return;
    const int rowb = p.S * 4;                                   // bytes of a value row
    const int vcap = C::VAL_BYTES / rowb - 4;                   // edges a value buffer holds (minus the alignment slack)
    // X areas zeroed once: chunks at or beyond Fin are never written by a DMA
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
        for (int i = tid; i < C::X_BYTES / 16; i += C::NT)
            *reinterpret_cast<f32x4*>(lds_raw + C::OFF_SLOT + sl * C::SLOT_BYTES + C::OFF_X + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // A group is worked off in row blocks of RB = 128 >> sh rows (all S supports of a block's edges in one value buffer):
    // sh from the group's edge count alone, so that loaders and compute waves agree without talking
    struct Geo { int kb, ne, kb4, ne4, lo8, nwin8, sh; bool staged; };
    auto geo_of = [&](int g) -> Geo {
        const int4 v = *reinterpret_cast<const int4*>(lds_raw + C::OFF_REC + (g & 3) * 256);
        Geo q;
        q.kb = __builtin_amdgcn_readfirstlane(v.x); q.ne = __builtin_amdgcn_readfirstlane(v.y);
        const int lo = __builtin_amdgcn_readfirstlane(v.z), nwin = __builtin_amdgcn_readfirstlane(v.w);
        q.kb4 = q.kb & ~3; q.ne4 = q.ne + (q.kb & 3);
        q.lo8 = lo & ~(C::XRB - 1); q.nwin8 = q.ne > 0 ? lo + nwin - q.lo8 : 0;      // (block-aligned window start)
        q.staged = q.ne4 <= ECAP && q.nwin8 <= XCAP;
        int sh = 0;
        while (sh < (W48 ? 4 : 3) && ((q.ne * 5 / 4) >> sh) + 8 > vcap) ++sh;   // 25 % headroom for uneven blocks (W48: smaller value buffers, blocks down to 8 rows)
        q.sh = sh;
        return q;
    };
    // edges [ks, ke) of row block b of group g (row pointers from the group's slot unless the block is the whole group)
    auto block_edges = [&](int g, const Geo& q, int b, int& ks, int& ke) {
        if (q.sh == 0) { ks = q.kb; ke = q.kb + q.ne; return; }
        const int* rp_l = reinterpret_cast<const int*>(lds_raw + C::OFF_SLOT + (g & 1) * C::SLOT_BYTES);
        const int rb = ROWS >> q.sh;
        const int nr = (int)min((int64_t)ROWS, p.nrows - (int64_t)g * ROWS);
        ks = __builtin_amdgcn_readfirstlane(rp_l[min(b * rb, nr)]);
        ke = __builtin_amdgcn_readfirstlane(rp_l[min((b + 1) * rb, nr)]);
    };

    if (wave >= 8) {
        // ===================================================== loader waves
        const int li = wave - 8;
        const int etot = p.rowptr[p.nrows];
        const uint32_t lds0 = (uint32_t)(uintptr_t)((gml_lds_void*)lds_raw);
        const u32x4 rs_rec = gml_raw_rsrc(p.ginfo, (uint32_t)p.ngroups * (GML_GREC_INTS(128) * 4));
        const u32x4 rs_rp = gml_raw_rsrc(p.rowptr, (uint32_t)(p.nrows + 1) * 4u);
        const u32x4 rs_x = gml_raw_rsrc(p.x, (uint32_t)(p.nrows * p.ldx) * 4u);
        const int ldxb = (int)p.ldx * 4;
        auto issue_group = [&](int g, const Geo& q) {           // row pointers, column ids, X window of group g; record of g + 2
            const uint32_t slot = lds0 + C::OFF_SLOT + (g & 1) * C::SLOT_BYTES;
            if (q.staged) {
                const u32x4 rs_col = gml_raw_rsrc(p.col + q.kb4, (uint32_t)min((uint64_t)(etot - q.kb4) * 4u, (uint64_t)0xffffff00u));
                if constexpr (!W48) {
                    const int nxi = (q.nwin8 + 7) >> 3;
                    const int c4 = (lane & 7) * 4;
                    for (int i = li; i < nxi; i += NL) {
                        const int rr = q.lo8 + 8 * i + (lane >> 3);
                        if (c4 < p.Fin) gml_dma16(rs_x, slot + C::OFF_X + i * C::XBLK, rr * ldxb + c4 * 4);
                    }
                } else {                                        // 16 rows x 12 pieces of 16 bytes = three instructions per block
                    const int nxi = (q.nwin8 + 15) >> 4;
                    for (int i = li; i < nxi; i += NL) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const int qd = 64 * j + lane;
                            const int row = (qd * 171) >> 11, ch = qd - row * 12;      // / 12 (exact for qd < 192)
                            if (ch * 4 < p.Fin) gml_dma16(rs_x, slot + C::OFF_X + i * C::XBLK + j * 1024, (q.lo8 + 16 * i + row) * ldxb + ch * 16);
                        }
                    }
                }
                const int nci = (q.ne4 + 255) >> 8;
                for (int j = (li + 1) % NL; j < nci; j += NL)
                    if (256 * j + 4 * lane < q.ne4) gml_dma16(rs_col, slot + C::OFF_COL + j * 1024, (256 * j + 4 * lane) * 4);
            }
            if (li == NL - 1 && lane < 33) gml_dma16(rs_rp, slot, (g * ROWS + 4 * lane) * 4);
            if (li == NL - 2 && lane < 9 && g + 2 < g1)
                gml_dma16(rs_rec, lds0 + C::OFF_REC + ((g + 2) & 3) * 256, (g + 2) * (GML_GREC_INTS(128) * 4) + lane * 16);
        };
        auto issue_val = [&](int it, int g, const Geo& q, int b) {   // whole value rows of the block's edges, contiguous
            if (!q.staged) return;
            int ks = q.kb, ke = q.kb + q.ne;
            if (q.sh != 0) {                                    // block boundaries straight from global memory (scalar loads): the
                const int rb = ROWS >> q.sh;                    // group's row pointers may still be on their way into LDS
                const int64_t r0 = (int64_t)g * ROWS;
                ks = p.rowptr[min(r0 + b * rb, p.nrows)];
                ke = p.rowptr[min(r0 + (b + 1) * rb, p.nrows)];
            }
            const int ks4 = ks & ~3, n = ke - ks4;
            if (n > vcap + 3) return;                           // (the compute waves see the same count and gather from global)
            const uint32_t dst = lds0 + C::OFF_VAL + (it & 1) * C::VAL_BYTES;
            const uint64_t left = (uint64_t)(etot - ks4) * rowb;
            const u32x4 rs_val = gml_raw_rsrc(p.val + (int64_t)ks4 * p.S, (uint32_t)min(left, (uint64_t)0xffffff00u));
            const int nb = n * rowb;
            for (int j = li; j * 1024 < nb; j += NL)
                if (j * 1024 + lane * 16 < nb) gml_dma16(rs_val, dst + j * 1024, j * 1024 + lane * 16);
        };
        if (li == NL - 2 && lane < 9) {
            gml_dma16(rs_rec, lds0 + C::OFF_REC + (g0 & 3) * 256, g0 * (GML_GREC_INTS(128) * 4) + lane * 16);
            if (g0 + 1 < g1) gml_dma16(rs_rec, lds0 + C::OFF_REC + ((g0 + 1) & 3) * 256, (g0 + 1) * (GML_GREC_INTS(128) * 4) + lane * 16);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // (A) records landed, X areas zeroed
        asm volatile("" ::: "memory");
        issue_group(g0, geo_of(g0));
        issue_val(0, g0, geo_of(g0), 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int it = 0;
        for (int g = g0; g < g1; ++g) {
            const Geo q = geo_of(g);
            const int nblk = 1 << q.sh;
            for (int b = 0; b < nblk; ++b, ++it) {
                __builtin_amdgcn_s_barrier();                   // (B) item complete in LDS; every wave has left the previous one
                asm volatile("" ::: "memory");
                // next group's data as early as possible
                if (b == 0 && g + 1 < g1) issue_group(g + 1, geo_of(g + 1));
                if (b + 1 < nblk) issue_val(it + 1, g, q, b + 1);
                else if (g + 1 < g1) issue_val(it + 1, g + 1, geo_of(g + 1), 0);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
        }
        return;
    }

    // ========================================================= compute waves
    // lane = (r8 = lane >> 3, j = lane & 7): 8 rows per wave, the 8 lanes of a row keep features 4 j .. 4 j + 3 of 8 supports
    // each -- so that ONE 16-byte store instruction writes whole 128-byte (row, support) lines (8 adjacent lanes per line; with
    // a row's lanes 16 apart, as the MFMA-shaped kernels have them, every instruction wrote half lines: 0.55 instead of 0.75 of
    // the roof at ZINC shapes)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // (A)
    asm volatile("" ::: "memory");
    // lanes <-> (row of the tile, 4 features): 8 rows x 8 feature quads; a feature chunk of <= 16 (the second launch of a 48-feature
    // input) would leave half the lanes without a feature: 16 rows x 4 quads then (round 4: Fin = 48 SpMM 0.41 -> see DESIGN s5)
    const bool narrow = !W48 && p.Fin <= 16;
    const int tr = narrow ? 16 : 8;                            // rows of a tile
    const int r8 = narrow ? lane >> 2 : lane >> 3, j4 = narrow ? (lane & 3) * 4 : (lane & 7) * 4;
    const bool fok = j4 < p.Fin;
    const int64_t hrow = (int64_t)p.S * p.hs;                  // floats between rows of H
    const int nchunk = (p.S + 7) >> 3;
    int it = 0;
    for (int g = g0; g < g1; ++g) {
        int nblk = 1;
        for (int b = 0; b < nblk; ++b, ++it) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // (B)
            asm volatile("" ::: "memory");
            const Geo q = geo_of(g);
            nblk = 1 << q.sh;
            const unsigned char* slot = lds_raw + C::OFF_SLOT + (g & 1) * C::SLOT_BYTES;
            const int* rp_l = reinterpret_cast<const int*>(slot);
            const int* col_l = reinterpret_cast<const int*>(slot + C::OFF_COL);
            const float* ea_l = reinterpret_cast<const float*>(lds_raw + C::OFF_VAL + (it & 1) * C::VAL_BYTES);
            const int64_t r0 = (int64_t)g * ROWS;
            const int nr = (int)min((int64_t)ROWS, p.nrows - r0);
            int ks, ke_b;
            block_edges(g, q, b, ks, ke_b);
            const int ks4 = ks & ~3;
            const bool staged = q.staged && (ke_b - ks4) <= vcap + 3;
            const int rb = ROWS >> q.sh, ntile = rb / tr;        // (rb >= tr: 16-row tiles only without W48, whose sh <= 3)
            // byte offset of (window row c, features j4 ..): xoff + XROW c + 16 (c / XRB)
            const int xoff = C::OFF_SLOT + (g & 1) * C::SLOT_BYTES + C::OFF_X + j4 * 4 - q.lo8 * C::XROW - (q.lo8 / C::XRB) * 16;
            const auto hrs = __builtin_amdgcn_make_buffer_rsrc(p.h + r0 * hrow, 0, (int)(nr * hrow * 4), 0x00020000);
            // work units (8-row tile, chunk of 8 supports) dealt to the waves
            for (int u = wave; u < ntile * nchunk; u += 8) {
                const int tile = u % ntile, ch = u / ntile;
                const int row = b * rb + tile * tr + r8;
                const bool rvalid = row < nr;
                const int kbeg = rvalid ? rp_l[row] : 0;
                const int kend = rvalid ? rp_l[row + 1] : 0;
                constexpr int NQ = W48 ? 2 : 1;                 // feature quads per lane: 4 j .. 4 j + 3 and (W48, lanes j < 4) 32 + 4 j .. + 3
                f32x2 acc[8][2 * NQ];
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int h = 0; h < 2 * NQ; ++h) acc[s][h] = f32x2{0.f, 0.f};
                if (staged) {
                    int k = kbeg;
                    const int ke = kend;
                    if (k < ke) {
                        struct Ops { f32x4 e0, e1, t, t2; };
                        auto fetch = [&](Ops& o, int kk, int c) {
                            const float* er = ea_l + (kk - ks4) * p.S + 8 * ch;
                            if constexpr (VA == 4) {
                                o.e0 = *reinterpret_cast<const f32x4*>(er);
                                o.e1 = *reinterpret_cast<const f32x4*>(er + 4);
                            } else if constexpr (VA == 2) {
                                const f32x2 a0 = *reinterpret_cast<const f32x2*>(er), a1 = *reinterpret_cast<const f32x2*>(er + 2);
                                const f32x2 a2 = *reinterpret_cast<const f32x2*>(er + 4), a3 = *reinterpret_cast<const f32x2*>(er + 6);
                                o.e0 = f32x4{a0.x, a0.y, a1.x, a1.y}; o.e1 = f32x4{a2.x, a2.y, a3.x, a3.y};
                            } else {
                                o.e0 = f32x4{er[0], er[1], er[2], er[3]}; o.e1 = f32x4{er[4], er[5], er[6], er[7]};
                            }
                            const int xa = xoff + c * C::XROW + ((c / C::XRB) << 4);
                            o.t = *reinterpret_cast<const f32x4*>(lds_raw + xa);
                            if constexpr (W48) o.t2 = *reinterpret_cast<const f32x4*>(lds_raw + xa + 128 - ((lane & 4) << 4));   // (lanes j >= 4: a readable address, result unused)
                        };
                        auto fma = [&](const Ops& o) {
                            const f32x2 x0 = f32x2{o.t.x, o.t.y}, x1 = f32x2{o.t.z, o.t.w};
#pragma unroll
                            for (int s = 0; s < 8; ++s) {
                                const float ev = s < 4 ? o.e0[s & 3] : o.e1[s & 3];
                                const f32x2 e2 = f32x2{ev, ev};
                                acc[s][0] = e2 * x0 + acc[s][0];
                                acc[s][1] = e2 * x1 + acc[s][1];
                                if constexpr (W48) {
                                    acc[s][2] = e2 * f32x2{o.t2.x, o.t2.y} + acc[s][2];
                                    acc[s][3] = e2 * f32x2{o.t2.z, o.t2.w} + acc[s][3];
                                }
                            }
                        };
                        Ops A, B;
                        const int klast = ke - 1;
                        fetch(A, k, col_l[k - q.kb4]);
                        int cn = col_l[min(k + 1, klast) - q.kb4];
                        for (;;) {
                            const int c2 = col_l[min(k + 2, klast) - q.kb4];
                            fetch(B, min(k + 1, klast), cn);
                            fma(A);
                            if (++k >= ke) break;
                            cn = col_l[min(k + 2, klast) - q.kb4];
                            fetch(A, min(k + 1, klast), c2);
                            fma(B);
                            if (++k >= ke) break;
                        }
                    }
                } else {                                        // block outside the LDS capacities: global gathers
                    for (int k = kbeg; k < kend; ++k) {
                        const int src = p.col[k];
                        const float* xr = p.x + (int64_t)src * p.ldx + j4;
                        float xb[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) xb[t] = fok ? xr[t] : 0.f;
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            const float e = (8 * ch + s < p.S) ? p.val[(int64_t)k * p.S + 8 * ch + s] : 0.f;
                            const f32x2 e2 = f32x2{e, e};
                            acc[s][0] = e2 * f32x2{xb[0], xb[1]} + acc[s][0];
                            acc[s][1] = e2 * f32x2{xb[2], xb[3]} + acc[s][1];
                        }
                        if constexpr (W48) {
                            float xc[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) xc[t] = (32 + j4 + t < p.Fin && j4 < 16) ? xr[32 + t] : 0.f;
#pragma unroll
                            for (int s = 0; s < 8; ++s) {
                                const float e = (8 * ch + s < p.S) ? p.val[(int64_t)k * p.S + 8 * ch + s] : 0.f;
                                const f32x2 e2 = f32x2{e, e};
                                acc[s][2] = e2 * f32x2{xc[0], xc[1]} + acc[s][2];
                                acc[s][3] = e2 * f32x2{xc[2], xc[3]} + acc[s][3];
                            }
                        }
                    }
                }
                // stores: H[r0 + row][8 ch + s][hf0 + 4 j ..]: 8 lanes = one 128-byte line; lanes without a row / feature /
                // support get an out-of-range offset (dropped by the hardware): unpredicated, non-temporal
                const int base = (row * (int)hrow + 8 * ch * p.hs + p.hf0 + j4) * 4;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const bool ok = rvalid && fok && 8 * ch + s < p.S;
                    const u32x4 a = u32x4{__float_as_uint(acc[s][0].x), __float_as_uint(acc[s][0].y), __float_as_uint(acc[s][1].x), __float_as_uint(acc[s][1].y)};
                    __builtin_amdgcn_raw_buffer_store_b128(a, hrs, ok ? base + s * p.hs * 4 : 0x7fffff00, 0, /*nt*/ 2);
                    if constexpr (W48) {
                        const bool ok2 = rvalid && j4 < 16 && 32 + j4 < p.Fin && 8 * ch + s < p.S;
                        const u32x4 a2 = u32x4{__float_as_uint(acc[s][2].x), __float_as_uint(acc[s][2].y), __float_as_uint(acc[s][3].x), __float_as_uint(acc[s][3].y)};
                        __builtin_amdgcn_raw_buffer_store_b128(a2, hrs, ok2 ? base + 128 + s * p.hs * 4 : 0x7fffff00, 0, /*nt*/ 2);
                    }
                }
            }
        }
    }
}

template <int VA, bool W48 = false>
static int gml_launch_spmm3(const GmlSpmm3Params& p, dim3 grid, hipStream_t st) {
    GML_ALLOW_BIG_LDS(rc_, (&gml_k_spmm3<VA, W48>), 160 * 1024)
    if (rc_ != hipSuccess) return (int)rc_;
    hipLaunchKernelGGL((gml_k_spmm3<VA, W48>), grid, dim3(GmlSpmm3Cfg::NT), GmlSpmm3CfgT<W48>::lds_bytes(), st, p);
    return gml_launch_status();
}
