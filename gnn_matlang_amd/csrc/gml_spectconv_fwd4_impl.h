// Fused forward on the LDS-DMA landing ring, generalised (round 4): any row degree, S in {4, 6, 8}, Fin <= 48, Fout <= 32
//
//   out[r, :] = act( sum_s (sum_{k in row r} val[pos(k), s] x[col[k], :]) W_s + b )      libs/spect_conv.py:76-80, :93-94
//
// Same machine as gml_k_spectconv_fwd3 (gml_spectconv_fwd3_impl.h: 8 compute waves + 4 loader waves, everything a group needs
// lands in LDS by `buffer_load ... lds`, one barrier per work item, compute waves never issue a load).  What is new:
//   * WORK ITEMS ARE (group, edge chunk).  A 128-row group with more edges than one edge buffer holds (sr25.py's supports have
//     13 entries per row = 1,664 per group; fwd3 sent such groups to global gathers: 0.10 of the HBM roof) is walked in
//     balanced chunks: the loaders fill the OTHER edge buffer with the next chunk while the compute waves aggregate the current
//     one, the accumulators live across the chunks of a group, projection and stores happen after its last chunk.  The chunks
//     are derived from the group record (first edge, edge count) by every wave on its own: no new record format.
//   * The landing zones are split by lifetime: two GROUP buffers (row pointers + X window, alternating per group) and two EDGE
//     buffers (column ids + value rows, alternating per item), so a group's X window is fetched once however many chunks it has.
//   * FB = 1: 48 input features (sr25.py:252-262, mutag.py:272-288: hidden width 32 + 16 / 24 + 24).  A lane keeps features
//     8 kq .. 8 kq + 7 and 32 + 4 kq .. + 3 of its row (12 per support); the projection adds one K = 16 MFMA triple per support
//     and column block (v_mfma_f32_16x16x16_bf16) against a second W image [s][o][16 f].  X rows are 192 bytes: sixteen of them
//     are exactly three DMA instructions; blocks 3088 bytes apart.
//   * S = 6 (sr25): 24-byte value rows.  Stored in CSR order they are one contiguous byte range per chunk (16-byte pieces, no
//     row structure needed); gathered through a position map (EP) they land with `buffer_load_dwordx3 ... lds`, two lanes per
//     row, 12 bytes per lane (probed: tools/probes/probe_glds3.hip).
// Groups whose column window exceeds the staged rows take the global-gather path (same results), as in fwd3.
#pragma once
#include "gml_common.h"
#include "gml_spectconv_impl.h"
#include "gml_spectconv_fwd3_impl.h"

typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

// 12 bytes per active lane from rs[voff] to LDS byte address lds_addr + 12 * lane (see gml_dma16 for the statement's shape)
__device__ __forceinline__ void gml_dma12(u32x4 rs, uint32_t lds_addr, int voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx3 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}

#define GML_FWD4_NT 768                                        // 8 compute waves + 4 loader waves: 3 waves per SIMD

template <int S, int FB, bool EP>
struct GmlFwd4Cfg {
    static constexpr int ROWS = 128;
    static constexpr int NLOAD = 4;
    static constexpr int NT = GML_FWD4_NT;
    static_assert(NT == 512 + 64 * NLOAD, "waves");
    static constexpr int FR = 32 + 16 * FB;                    // floats of a staged x row
    static constexpr int XRB = FB ? 16 : 8;                    // rows per window block (a whole number of 1 KiB DMA instructions)
    static constexpr int XIB = XRB * FR * 4 / 1024;            // DMA instructions per block (1 or 3)
    static constexpr int XBLK = XRB * FR * 4 + 16;             // bytes between blocks: the 16-byte pad spreads the b128 gathers over the bank groups
    static constexpr int XNBLK = FB ? 13 : 25;
    static constexpr int XCAP = XNBLK * XRB;                   // staged window rows incl. alignment slack (208 / 200)
    static constexpr int X_BYTES = XNBLK * XBLK;
    static constexpr int W_HALF = S * 32 * 32;                 // bf16 elements of one (hi or lo) W image [s][o][32 f]
    static constexpr int W2_HALF = FB ? S * 32 * 16 : 0;       // second image [s][o][16 f]: features 32 .. 47
    static constexpr int W_BYTES = 4 * (W_HALF + W2_HALF);
    static constexpr int REC_BYTES = 4 * 256;                  // ring of 4 group records
    static constexpr int RP_BYTES = 528;                       // 132 row pointers
    static constexpr int GRP_BYTES = RP_BYTES + X_BYTES;       // group buffer: row pointers + X window
    static constexpr int VROW = 4 * S;
    static constexpr int AVAIL = 160 * 1024 - W_BYTES - REC_BYTES - 2 * GRP_BYTES;
    static constexpr int ECAP_RAW = AVAIL / (2 * (4 + VROW) + (EP ? 8 : 0));
    static constexpr int ECAP = ECAP_RAW >= 1024 ? 1024 : ECAP_RAW / 64 * 64;      // staged edges per item
    static constexpr int COL_BYTES = ECAP * 4, VAL_BYTES = ECAP * VROW;
    static constexpr int EDGE_BYTES = COL_BYTES + VAL_BYTES;
    static constexpr int OFF_W2 = 4 * W_HALF;                  // (hi, lo of the first image, then hi, lo of the second)
    static constexpr int OFF_REC = W_BYTES, OFF_GRP = OFF_REC + REC_BYTES, OFF_EPOS = OFF_GRP + 2 * GRP_BYTES;
    static constexpr int OFF_EDGE = OFF_EPOS + (EP ? 2 * COL_BYTES : 0);
    static constexpr int OFF_X = RP_BYTES;                     // inside a group buffer
    static constexpr int OFF_VAL = COL_BYTES;                  // inside an edge buffer
    static constexpr size_t lds_bytes() { return (size_t)OFF_EDGE + 2 * (size_t)EDGE_BYTES; }
    static_assert(ECAP >= 256, "edge buffers too small");
    static_assert(GRP_BYTES % 16 == 0 && EDGE_BYTES % 16 == 0 && OFF_EDGE % 16 == 0 && OFF_GRP % 16 == 0, "16-byte aligned landing zones");
    static_assert(OFF_EDGE + 2 * EDGE_BYTES <= 160 * 1024, "LDS budget");
};

template <int S, int FB, int NOB, bool EP>
__global__ __launch_bounds__(GML_FWD4_NT, 1) void gml_k_spectconv_fwd4(const GmlFwdParams p) {
    using C = GmlFwd4Cfg<S, FB, EP>;
    static_assert(S % 2 == 0, "value rows are read as float2 / float4");
    constexpr int ROWS = C::ROWS, ECAP = C::ECAP, XCAP = C::XCAP, VROW = C::VROW, FR = C::FR;
    constexpr int VAL_ALIGN = (S % 4 == 0) ? 4 : 2;
    constexpr int NH = 4 + 2 * FB;                             // f32x2 accumulators per support
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* Wof_h = reinterpret_cast<__bf16*>(lds_raw);       // [s][o][32 f], 16-byte chunks XOR-swizzled by gml_wkey(o)
    __bf16* Wof_l = Wof_h + C::W_HALF;
    __bf16* W2_h = reinterpret_cast<__bf16*>(lds_raw + C::OFF_W2);   // [s][o][16 f] (features 32 + f), 32-byte rows
    __bf16* W2_l = W2_h + C::W2_HALF;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);
    if (g0 >= g1) return;
    const bool loader = wave >= 8;

    // ---- once per workgroup: W images, zeroed X areas (chunks at or beyond Fin are never written by a DMA and stay zero)
    for (int e = tid; e < S * 32 * 32; e += C::NT) {
        const int f = e & 31, o = (e >> 5) & 31, s = e >> 10;
        const float v = (f < p.Fin && o < p.Fout) ? p.w[(int64_t)(p.s0 + s) * p.w_ss + (int64_t)f * p.w_si + (int64_t)o * p.w_so] : 0.f;
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        const int iof = (s * 32 + o) * 32 + ((((f >> 3) ^ gml_wkey(o)) & 3) << 3) + (f & 7);
        Wof_h[iof] = h;
        Wof_l[iof] = l;
    }
    if constexpr (FB) {
        for (int e = tid; e < S * 32 * 16; e += C::NT) {
            const int f2 = e & 15, o = (e >> 4) & 31, s = e >> 9;
            const int f = 32 + f2;
            const float v = (f < p.Fin && o < p.Fout) ? p.w[(int64_t)(p.s0 + s) * p.w_ss + (int64_t)f * p.w_si + (int64_t)o * p.w_so] : 0.f;
            const __bf16 h = (__bf16)v;
            const __bf16 l = (__bf16)(v - (float)h);
            W2_h[e] = h;
            W2_l[e] = l;
        }
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
        for (int i = tid; i < C::X_BYTES / 16; i += C::NT)
            *reinterpret_cast<f32x4*>(lds_raw + C::OFF_GRP + sl * C::GRP_BYTES + C::OFF_X + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // a group's geometry from its record in the ring (wave-uniform): edges kb4 .. kb4 + ne4 (16-byte aligned start), window rows
    // lo_a .. lo_a + nwin (block aligned start), nch chunks of cs edges
    struct Geo { int kb4, ne4, lo_a, nwin, nch, cs; bool staged; };
    const int etot = p.rowptr[p.nrows];
    const bool bigv = (uint64_t)etot * VROW > 0xffffff00ull;  // value rows beyond 32-bit byte offsets: every group gathers from global memory
    auto geo_of = [&](int g) -> Geo {
        const int4 v = *reinterpret_cast<const int4*>(lds_raw + C::OFF_REC + (g & 3) * 256);
        const int kb = __builtin_amdgcn_readfirstlane(v.x), ne = __builtin_amdgcn_readfirstlane(v.y);
        const int lo = __builtin_amdgcn_readfirstlane(v.z), nwin = __builtin_amdgcn_readfirstlane(v.w);
        const int r0 = g * ROWS;
        const int wlo = ne > 0 ? lo : r0, whi = ne > 0 ? lo + nwin : r0;
        Geo q;
        q.kb4 = kb & ~3; q.ne4 = ne + (kb & 3);
        q.lo_a = wlo & ~(C::XRB - 1); q.nwin = whi - q.lo_a;
        q.staged = q.nwin <= XCAP && !bigv;
        q.nch = (q.staged && q.ne4 > ECAP) ? (q.ne4 + ECAP - 1) / ECAP : 1;
        q.cs = q.nch == 1 ? max(q.ne4, 4) : (((q.ne4 + q.nch - 1) / q.nch + 3) & ~3);
        return q;
    };

    if (loader) {
        // =====================================================================================================
        // Loader waves: all LDS-DMA of the workgroup.  Trip of item i: (barrier) -> edges of item i + 1 into the other edge buffer
        // (+ row pointers and X window when it opens a new group), value positions of item i + 2, record of group g + 3 ->
        // wait until everything has landed -> (next barrier).
        // =====================================================================================================
        const uint32_t lds0 = (uint32_t)(uintptr_t)((gml_lds_void*)lds_raw);
        const u32x4 rs_rec = gml_raw_rsrc(p.ginfo, (uint32_t)p.ngroups * (GML_GREC_INTS(128) * 4));
        const u32x4 rs_rp = gml_raw_rsrc(p.rowptr, (uint32_t)(p.nrows + 1) * 4u);
        const u32x4 rs_col = gml_raw_rsrc(p.col, bigv ? 0u : (uint32_t)etot * 4u);
        const u32x4 rs_val = gml_raw_rsrc(p.val, bigv ? 0u : (uint32_t)etot * VROW);
        const u32x4 rs_epos = gml_raw_rsrc(p.epos, (EP && !bigv) ? (uint32_t)etot * 4u : 0u);
        const u32x4 rs_x = gml_raw_rsrc(p.x, (uint32_t)(p.nrows * p.ldx) * 4u);
        const int ldxb = (int)p.ldx * 4;
        constexpr int NL = C::NLOAD;
        const int li = wave - 8;
        // X window: per-lane source offsets of the XIB instructions of a block (row inside the block, 16-byte chunk of the row)
        int xsrc[C::XIB];
        bool xon[C::XIB];
#pragma unroll
        for (int j = 0; j < C::XIB; ++j) {
            const int qd = 64 * j + lane;                      // 16-byte piece of the block
            const int row = FB ? (qd * 171) >> 11 : qd >> 3;   // / 12 (exact for qd < 192) or / 8
            const int ch = qd - row * (FR / 4);
            xsrc[j] = row * ldxb + ch * 16;
            xon[j] = ch * 4 < p.Fin;
        }
        auto dma_rec = [&](int g) {                            // record of group g -> ring entry g & 3
            if (li == NL - 1 && lane < 9) gml_dma16(rs_rec, lds0 + C::OFF_REC + (g & 3) * 256, g * (GML_GREC_INTS(128) * 4) + lane * 16);
        };
        auto dma_epos = [&](int it, int e0, int n) {           // value positions of an item (256 per instruction) -> position buffer it & 1
            const int nci = (n + 255) >> 8;
            for (int j = li; j < nci; j += NL)
                if (256 * j + 4 * lane < n) gml_dma16(rs_epos, lds0 + C::OFF_EPOS + (it & 1) * C::COL_BYTES + j * 1024, (e0 + 256 * j + 4 * lane) * 4);
        };
        auto chunk_of = [&](const Geo& q, int c, int& e0, int& n) {
            e0 = q.kb4 + c * q.cs;
            n = min(q.cs, q.ne4 - c * q.cs);
        };
        // the item after (g, c); false at the end of this workgroup's range
        auto next_of = [&](int g, int c, const Geo& q, int& ng, int& nc, Geo& nq) -> bool {
            ng = g; nc = c + 1; nq = q;
            if (nc < q.nch) return true;
            nc = 0; ng = g + 1;
            if (ng >= g1) return false;
            nq = geo_of(ng);
            return true;
        };
        auto issue = [&](int it, int g, int c, const Geo& q) {
            const uint32_t ebuf = lds0 + C::OFF_EDGE + (it & 1) * C::EDGE_BYTES;
            const uint32_t gbuf = lds0 + C::OFF_GRP + (g & 1) * C::GRP_BYTES;
            if (q.staged) {
                int e0, n;
                chunk_of(q, c, e0, n);
                // ---- value rows
                if constexpr (!EP) {
                    // CSR order: one contiguous byte range, 16-byte pieces (any S)
                    const int npc = (n * VROW + 15) >> 4;
                    for (int j = li; 64 * j < npc; j += NL)
                        if (64 * j + lane < npc) gml_dma16(rs_val, ebuf + C::OFF_VAL + j * 1024, e0 * VROW + (64 * j + lane) * 16);
                } else {
                    const int* epos_l = reinterpret_cast<const int*>(lds_raw + C::OFF_EPOS + (it & 1) * C::COL_BYTES);
                    constexpr int PB = (S % 4 == 0) ? 16 : 12;             // bytes per lane
                    constexpr int LPE = VROW / PB;                         // lanes per value row
                    constexpr int EPI = 64 / LPE;                          // value rows per instruction
                    const int nvi = (n + EPI - 1) / EPI;
                    const int part = (lane % LPE) * PB;
                    for (int j0 = 4 * li; j0 < nvi; j0 += 4 * NL) {        // four position reads, then their four gathers
                        int voff[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int e = min((j0 + u) * EPI + lane / LPE, ECAP - 1);
                            voff[u] = epos_l[e] * VROW + part;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int e = (j0 + u) * EPI + lane / LPE;
                            if (e < n) {
                                if constexpr (PB == 16) gml_dma16(rs_val, ebuf + C::OFF_VAL + (j0 + u) * 1024, voff[u]);
                                else gml_dma12(rs_val, ebuf + C::OFF_VAL + (j0 + u) * 768, voff[u]);
                            }
                        }
                    }
                }
                // ---- column ids: 256 per instruction
                const int nci = (n + 255) >> 8;
                for (int j = (li + 2) % NL; j < nci; j += NL)
                    if (256 * j + 4 * lane < n) gml_dma16(rs_col, ebuf + j * 1024, (e0 + 256 * j + 4 * lane) * 4);
                // ---- a new group: its X window, whole blocks
                if (c == 0) {
                    const int nblk = (q.nwin + C::XRB - 1) / C::XRB;
                    for (int b = NL - 1 - li; b < nblk; b += NL) {         // (the loaders with fewer value batches first)
                        const int xo = (q.lo_a + C::XRB * b) * ldxb;
                        const uint32_t xd = gbuf + C::OFF_X + b * C::XBLK;
#pragma unroll
                        for (int j = 0; j < C::XIB; ++j)
                            if (xon[j]) gml_dma16(rs_x, xd + j * 1024, xo + xsrc[j]);
                    }
                }
            }
            if (c == 0 && li == NL - 1 && lane < 33) gml_dma16(rs_rp, gbuf, (g * ROWS + 4 * lane) * 4);
            if constexpr (EP) {                                            // positions of the item after this one
                int ng, nc; Geo nq;
                if (next_of(g, c, q, ng, nc, nq) && nq.staged) {
                    int e0, n;
                    chunk_of(nq, nc, e0, n);
                    dma_epos(it + 1, e0, n);
                }
            }
        };

        dma_rec(g0);
        if (g0 + 1 < g1) dma_rec(g0 + 1);
        if (g0 + 2 < g1) dma_rec(g0 + 2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // (A) records landed, W images and zeroed X areas complete
        int g = g0, c = 0, it = 0;
        Geo q = geo_of(g0);
        if constexpr (EP) {
            if (q.staged) { int e0, n; chunk_of(q, 0, e0, n); dma_epos(0, e0, n); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // (A2) every loader's share of the first positions has landed
        }
        issue(0, g0, 0, q);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (;;) {
            __builtin_amdgcn_s_barrier();                      // (B) item `it` complete in its buffers; every wave has left the other ones
            asm volatile("" ::: "memory");
            if (c == 0 && g + 3 < g1) dma_rec(g + 3);          // (entry (g - 1) & 3: group g - 1 is finished)
            int ng, nc; Geo nq;
            const bool has = next_of(g, c, q, ng, nc, nq);
            if (has) issue(it + 1, ng, nc, nq);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (!has) break;
            g = ng; c = nc; q = nq; ++it;
        }
    } else {
        // =====================================================================================================
        // Compute waves: one 16-row tile each
        // =====================================================================================================
        float bias_r[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) bias_r[ob] = (p.bias && ob * 16 + r16 < p.Fout) ? p.bias[ob * 16 + r16] : 0.f;
        // one 16-byte store per lane and 16-column block when the row is written in whole float4 chunks (see fwd3)
        const int ncols = p.Fout;
        const bool wide = ncols % 4 == 0 && p.ldo % 4 == 0 && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // (A)
        if constexpr (EP) __builtin_amdgcn_s_barrier();        // (A2)

        // projection + stores of one group's tile: out tile = sum_s acc_s W_s (acc split on the fly = A fragments; K = 32: k = f =
        // 8 kq + j; K = 16: k = 4 kq + j <-> f = 32 + 4 kq + j)
        auto project = [&](f32x2 (&acc)[S][NH], int64_t r0, int nr, uint32_t out_rows) {
            f32x4 oacc[NOB];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) oacc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (FB: one fragment set -- 12 features per support leave no registers for a second one; three waves per SIMD cover the reads)
            constexpr int NST = FB ? 1 : 2;
            bf16x8 wh[NST][NOB], wl[NST][NOB];
            bf16x4v vh[NST][NOB], vl[NST][NOB];
            auto frag = [&](int s, int st) {
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) {
                    const int o = ob * 16 + r16;               // B[k = f][n = o]: 8 consecutive f of column o
                    const int off = (s * 32 + o) * 32 + (((kq ^ gml_wkey(o)) & 3) << 3);
                    wh[st][ob] = *reinterpret_cast<const bf16x8*>(Wof_h + off);
                    wl[st][ob] = *reinterpret_cast<const bf16x8*>(Wof_l + off);
                    if constexpr (FB) {
                        const int off2 = (s * 32 + o) * 16 + 4 * kq;
                        vh[st][ob] = *reinterpret_cast<const bf16x4v*>(W2_h + off2);
                        vl[st][ob] = *reinterpret_cast<const bf16x4v*>(W2_l + off2);
                    }
                }
            };
            if constexpr (NST == 2) frag(0, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int st = NST == 2 ? (s & 1) : 0;
                if constexpr (NST == 2) { if (s + 1 < S) frag(s + 1, st ^ 1); }
                else frag(s, 0);
                const float av[8] = {acc[s][0].x, acc[s][0].y, acc[s][1].x, acc[s][1].y, acc[s][2].x, acc[s][2].y, acc[s][3].x, acc[s][3].y};
                bf16x8 ah, al;
                gml_split8(av, ah, al);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oacc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[st][ob], oacc[ob], 0, 0, 0);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oacc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[st][ob], oacc[ob], 0, 0, 0);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oacc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[st][ob], oacc[ob], 0, 0, 0);
                if constexpr (FB) {
                    bf16x2 h0, l0, h1, l1;
                    gml_split2(acc[s][4].x, acc[s][4].y, h0, l0);
                    gml_split2(acc[s][5].x, acc[s][5].y, h1, l1);
                    const bf16x4v bh = bf16x4v{h0[0], h0[1], h1[0], h1[1]}, bl = bf16x4v{l0[0], l0[1], l1[0], l1[1]};
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) oacc[ob] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bl, vh[st][ob], oacc[ob], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) oacc[ob] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh, vl[st][ob], oacc[ob], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) oacc[ob] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh, vh[st][ob], oacc[ob], 0, 0, 0);
                }
            }
            // output stores through a buffer descriptor based at this group's first row: lanes outside (row >= nr, column >=
            // Fout) get an offset beyond the range and are dropped by the hardware -- no predicate
            const auto ors = __builtin_amdgcn_make_buffer_rsrc(p.out + r0 * p.ldo, 0, 0x7ffffe00, 0x00020000);
            const bool relu = (p.flags & GML_RELU) != 0;
            if (wide) {
                f32x4 ov[NOB];
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float v = oacc[ob][reg] + bias_r[ob];
                        ov[ob][reg] = relu ? fmaxf(v, 0.f) : v;
                    }
                const int lr = (int)((out_rows >> (8 * (r16 & 3))) & 255u);        // after the transpose: the row of register r16 & 3
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) {
                    gml_quad_transpose(ov[ob], lane);
                    const int cc = 16 * ob + 4 * (r16 >> 2);
                    const int off = (cc < ncols && lr < nr) ? (lr * (int)p.ldo + cc) * 4 : 0x7fffff00;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov[ob]), ors, off, 0, 0);
                }
            } else {
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) {
                    const int o = ob * 16 + r16;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int lr = (int)((out_rows >> (8 * reg)) & 255u);
                        float v = oacc[ob][reg] + bias_r[ob];
                        if (relu) v = fmaxf(v, 0.f);
                        const int off = (o < p.Fout && lr < nr) ? (lr * (int)p.ldo + o) * 4 : 0x7fffff00;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, off, 0, 0);
                    }
                }
            }
        };

        // One barrier per work item: a group's first item at the top of the group loop, its further chunks inside.  The accumulators
        // are scoped to ONE group (not carried around the group loop), and the global-gather road has its own set: with one set
        // shared by both roads the compiler kept two copies of it and moved between them (+48 registers, seen in the ISA).
        int it = 0;
        for (int g = g0; g < g1; ++g) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // (B) first item of group g
            asm volatile("" ::: "memory");
            const Geo q = geo_of(g);
            const unsigned char* gbuf = lds_raw + C::OFF_GRP + (g & 1) * C::GRP_BYTES;
            const int* rp_l = reinterpret_cast<const int*>(gbuf);
            const unsigned char* rec = lds_raw + C::OFF_REC + (g & 3) * 256;
            const int row = rec[16 + wave * 16 + r16];
            const uint32_t out_rows = reinterpret_cast<const uint32_t*>(rec + 16)[wave * 4 + kq];
            const int64_t r0 = (int64_t)g * ROWS;
            const int nr = (int)min((int64_t)ROWS, p.nrows - r0);
            const bool rvalid = row < nr;
            const int kbeg = rvalid ? rp_l[row] : 0;
            const int kend = rvalid ? rp_l[row + 1] : 0;

            if (!q.staged) {
                // ---- window outside the LDS capacity: global gathers (one item per group).  A wave-uniform loop (every lane runs the
                //      longest row's trip count, lanes past their row multiply by zero); chunks at or beyond Fin are read from the
                //      row's start instead and zeroed
                f32x2 acc[S][NH];
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int h = 0; h < NH; ++h) acc[s][h] = f32x2{0.f, 0.f};
                const bool on0 = 8 * kq < p.Fin, on1 = 8 * kq + 4 < p.Fin, on2 = FB && 32 + 4 * kq < p.Fin;
                const int o0 = on0 ? 8 * kq : 0, o1 = on1 ? 8 * kq + 4 : 0, o2 = on2 ? 32 + 4 * kq : 0;
                const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
                const int klim = max(etot, 1) - 1;
                for (int k = kbeg; __builtin_amdgcn_ballot_w64(k < kend) != 0ull; ++k) {
                    const bool on = k < kend;
                    const int kk = min(k, klim);
                    const float* xr = p.x + (int64_t)p.col[kk] * p.ldx;
                    const float* vr = p.val + (int64_t)(EP ? p.epos[kk] : kk) * p.S + p.s0;
                    float e[S];
                    gml_load_row<S, VAL_ALIGN>(vr, e);
                    f32x4 t[2 + FB];
                    t[0] = *reinterpret_cast<const f32x4*>(xr + o0);
                    t[1] = *reinterpret_cast<const f32x4*>(xr + o1);
                    if constexpr (FB) t[2] = *reinterpret_cast<const f32x4*>(xr + o2);
                    t[0] = on0 ? t[0] : zero4; t[1] = on1 ? t[1] : zero4;
                    if constexpr (FB) t[2] = on2 ? t[2] : zero4;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float ev = on ? e[s] : 0.f;
                        const f32x2 e2 = f32x2{ev, ev};
#pragma unroll
                        for (int h = 0; h < NH; ++h) acc[s][h] = e2 * f32x2{t[h >> 1][2 * (h & 1)], t[h >> 1][2 * (h & 1) + 1]} + acc[s][h];
                    }
                }
                project(acc, r0, nr, out_rows);
                ++it;
                continue;
            }

            // ---- aggregation (fp32 VALU, packed): acc[s][f] += val[k, s] * x[col[k], f], chunk by chunk
            // byte offset of (row cidx, features 8 kq ..) in the window: xoff + FR * 4 * cidx + 16 (cidx / XRB)
            const int xoff = C::OFF_GRP + (g & 1) * C::GRP_BYTES + C::OFF_X + kq * 32 - q.lo_a * (FR * 4) - (q.lo_a / C::XRB) * 16;
            f32x2 acc[S][NH];
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int h = 0; h < NH; ++h) acc[s][h] = f32x2{0.f, 0.f};
            for (int c = 0; c < q.nch; ++c, ++it) {
                if (c > 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();              // (B) further chunks of the group
                    asm volatile("" ::: "memory");
                }
                const unsigned char* ebuf = lds_raw + C::OFF_EDGE + (it & 1) * C::EDGE_BYTES;
                const int* col_l = reinterpret_cast<const int*>(ebuf);
                const float* ea_l = reinterpret_cast<const float*>(ebuf + C::OFF_VAL);
                const int e0 = q.kb4 + c * q.cs;
                const int e1 = min(e0 + q.cs, q.kb4 + q.ne4);
                int k = max(kbeg, e0) - e0;
                const int ke = min(kend, e1) - e0;
                if (k < ke) {
                    // software pipeline as in fwd3: operands of edge k + 1 and the column id of edge k + 2 are requested before
                    // the packed FMAs of edge k; two register sets, no rotation moves
                    struct Ops { float e[S]; f32x4 t0, t1, t2; };
                    auto fetch = [&](Ops& o, int kk, int cidx) {
                        gml_load_row<S, VAL_ALIGN>(ea_l + kk * S, o.e);
                        const int off = xoff + cidx * (FR * 4) + ((cidx / C::XRB) << 4);
                        o.t0 = *reinterpret_cast<const f32x4*>(lds_raw + off);
                        o.t1 = *reinterpret_cast<const f32x4*>(lds_raw + off + 16);
                        if constexpr (FB) o.t2 = *reinterpret_cast<const f32x4*>(lds_raw + off + 128 - 16 * kq);   // features 32 + 4 kq .. + 3
                    };
                    auto fma = [&](const Ops& o) {
                        f32x2 xv[NH];
                        xv[0] = f32x2{o.t0.x, o.t0.y}; xv[1] = f32x2{o.t0.z, o.t0.w}; xv[2] = f32x2{o.t1.x, o.t1.y}; xv[3] = f32x2{o.t1.z, o.t1.w};
                        if constexpr (FB) { xv[4] = f32x2{o.t2.x, o.t2.y}; xv[5] = f32x2{o.t2.z, o.t2.w}; }
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const f32x2 e2 = f32x2{o.e[s], o.e[s]};
#pragma unroll
                            for (int h = 0; h < NH; ++h) acc[s][h] = e2 * xv[h] + acc[s][h];
                        }
                    };
                    Ops A, B;
                    const int klast = ke - 1;
                    fetch(A, k, col_l[k]);
                    int cn = col_l[min(k + 1, klast)];
                    for (;;) {
                        const int c2 = col_l[min(k + 2, klast)];
                        fetch(B, min(k + 1, klast), cn);
                        fma(A);
                        if (++k >= ke) break;
                        cn = col_l[min(k + 2, klast)];
                        fetch(A, min(k + 1, klast), c2);
                        fma(B);
                        if (++k >= ke) break;
                    }
                }
            }
            project(acc, r0, nr, out_rows);
        }
    }
}

template <int S, int FB, int NOB>
int gml_launch_fwd4(const GmlFwdParams& p, dim3 grid, hipStream_t st);

#define GML_FWD4_LAUNCH(SV, FBV, NOBV, EPV)                                                                  \
    {                                                                                                        \
        GML_ALLOW_BIG_LDS(rc_, (&gml_k_spectconv_fwd4<SV, FBV, NOBV, EPV>), 160 * 1024)                      \
        if (rc_ != hipSuccess) return (int)rc_;                                                              \
        const size_t lds_ = GmlFwd4Cfg<SV, FBV, EPV>::lds_bytes();                                           \
        hipLaunchKernelGGL((gml_k_spectconv_fwd4<SV, FBV, NOBV, EPV>), grid, dim3(GML_FWD4_NT), lds_, st, p); \
        return gml_launch_status();                                                                          \
    }
#define GML_DEFINE_FWD4(SV, FBV, NOBV)                                                                       \
    template <>                                                                                              \
    int gml_launch_fwd4<SV, FBV, NOBV>(const GmlFwdParams& p, dim3 grid, hipStream_t st) {                   \
        if (p.epos != nullptr) GML_FWD4_LAUNCH(SV, FBV, NOBV, true)                                          \
        GML_FWD4_LAUNCH(SV, FBV, NOBV, false)                                                                \
    }
