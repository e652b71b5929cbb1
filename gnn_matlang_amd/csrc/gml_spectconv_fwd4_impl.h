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
//     row structure needed); gathered through a position map (EP) they land as two OVERLAPPING 16-byte lanes per row, bytes
//     0..15 and 8..23, in 32-byte LDS rows [v0 v1 v2 v3 | v2 v3 v4 v5] (`buffer_load_dwordx3 ... lds` was probed first: it
//     writes 12 bytes but strides 16 per lane, tools/probes/probe_glds3.hip -- no denser than this and a new instruction form).
// Groups whose column window exceeds the staged rows (208 / 200) are NOT served: gml_spectconv_fwd_stage_window() tells the
// caller, who keeps such batches on the 64-row family; the kernel marks their rows NaN rather than compute them wrongly.
#pragma once
#include "gml_common.h"
#include "gml_spectconv_impl.h"
#include "gml_spectconv_fwd3_impl.h"

typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
// K = 16 piece operands and their MFMA (features 32 .. 47), bf16 pairs or f16 pairs (GML_F16X3)
template <bool F16> struct GmlPiece4 { using T = bf16x4v; };
template <> struct GmlPiece4<true> { using T = f16x4v; };
__device__ __forceinline__ f32x4 gml_mfma_piece4(bf16x4v a, bf16x4v b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 gml_mfma_piece4(f16x4v a, f16x4v b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

// debugging builds (tools/build_variant.py <name> -DGML_F4DBG=<bits>; counters in p.prof, read by gml_debug_f4_counts):
//   8  after every item's barrier the compute waves compare what landed in LDS with global memory (0 value rows, 1 column ids,
//      2 X window, 3 row pointers, 4 checks done, 5 group record, 6 row mapping)
//   4  every landing zone (records, row pointers, X windows, edge buffers) starts as NaN patterns
//   64 the W images against global memory at the start and at the end of the kernel (0/1 start, 2/3 end)
#ifndef GML_F4DBG
#define GML_F4DBG 0
#endif
// delay experiments (-DGML_F4_DELAY=<bits>): ~8k idle cycles at 1 kernel start, 2 compute waves after barrier A, 4 compute waves after every
// item barrier, 8 before the projection, 16 loaders between their wait and the barrier
#ifndef GML_F4_DELAY
#define GML_F4_DELAY 0
#endif
#define GML_F4_IDLE() do { for (int z_ = 0; z_ < 64; ++z_) __builtin_amdgcn_s_sleep(2); } while (0)


#define GML_FWD4_NT 768                                        // 8 compute waves + 4 loader waves: 3 waves per SIMD

// X1: ONE X window instead of two (FB = 1 launches whose groups need chunks: the 40 KB saved double the edges of a work item -- sr25's
// 1,664-edge groups are 2 chunks instead of 4).  The next group's window then lands behind the current group's last chunk, beside its
// projection and stores: one more barrier per group (C), taken by every wave.
template <int S, int FB, bool EP, bool X1 = false>
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
    static constexpr int GRPS_BYTES = X1 ? 2 * RP_BYTES + X_BYTES : 2 * GRP_BYTES;   // all group buffers
    static constexpr int VROW = 4 * S;
    // bytes of a value row in LDS.  24-byte rows (S = 6) land as two OVERLAPPING 16-byte lanes per row, [v0 v1 v2 v3 | v2 v3 v4 v5]
    // in 32 bytes, whether they are copied in CSR order or gathered through a position map: every read of the aggregation is then
    // an aligned ds_read_b128.  (Packed 24-byte rows were read with ds_read2_b64 / ds_read_b64 -- and a wave's wait for a
    // ds_read2_b64 was seen to return before its second element had landed: wrong tiles on the first launches of a process,
    // DESIGN s4.1c.  The 12-byte LDS-DMA form strides 16 bytes per lane as well: tools/probes/probe_glds3.hip.)
    static constexpr int VROW_L = (S % 4 != 0) ? 32 : VROW;
    static_assert(S % 4 == 0 || S == 6, "value rows: multiples of 16 bytes, or 24 bytes");
    static constexpr int AVAIL = 160 * 1024 - W_BYTES - REC_BYTES - GRPS_BYTES;
    static constexpr int ECAP_RAW = AVAIL / (2 * (4 + VROW_L) + (EP ? 8 : 0));
    static constexpr int ECAP = ECAP_RAW >= 1024 ? 1024 : ECAP_RAW / 64 * 64;      // staged edges per item
    static constexpr int COL_BYTES = ECAP * 4, VAL_BYTES = ECAP * VROW_L;
    static constexpr int EDGE_BYTES = COL_BYTES + VAL_BYTES;
    static constexpr int OFF_W2 = 4 * W_HALF;                  // (hi, lo of the first image, then hi, lo of the second)
    static constexpr int OFF_REC = W_BYTES, OFF_GRP = OFF_REC + REC_BYTES, OFF_EPOS = OFF_GRP + GRPS_BYTES;
    static constexpr int OFF_EDGE = OFF_EPOS + (EP ? 2 * COL_BYTES : 0);
    // byte offsets of group g's row pointers / X window (two of each; X1: two row-pointer buffers, then the one window)
    __host__ __device__ static constexpr int rp_off(int g) { return OFF_GRP + (g & 1) * (X1 ? RP_BYTES : GRP_BYTES); }
    __host__ __device__ static constexpr int x_off(int g) { return X1 ? OFF_GRP + 2 * RP_BYTES : OFF_GRP + (g & 1) * GRP_BYTES + RP_BYTES; }
    static constexpr int OFF_VAL = COL_BYTES;                  // inside an edge buffer
    static constexpr size_t lds_bytes() { return (size_t)OFF_EDGE + 2 * (size_t)EDGE_BYTES; }
    static_assert(ECAP >= 256, "edge buffers too small");
    static_assert(GRP_BYTES % 16 == 0 && EDGE_BYTES % 16 == 0 && OFF_EDGE % 16 == 0 && OFF_GRP % 16 == 0, "16-byte aligned landing zones");
    static_assert(OFF_EDGE + 2 * EDGE_BYTES <= 160 * 1024, "LDS budget");
};

// F16 (round 6, GML_F16X3: see gml_spectconv_fwd3_impl.h): the projection on f16 (hi, lo) pieces under power-of-two scales
template <int S, int FB, int NOB, bool EP, bool X1 = false, bool F16 = false>
__global__ __launch_bounds__(GML_FWD4_NT, 1) void gml_k_spectconv_fwd4(const GmlFwdParams p) {
    using C = GmlFwd4Cfg<S, FB, EP, X1>;
    using FT = typename GmlPiece<F16>::T;
    using FT4 = typename GmlPiece4<F16>::T;
    static_assert(S % 2 == 0, "value rows are read as float2 / float4");
    constexpr int ROWS = C::ROWS, ECAP = C::ECAP, XCAP = C::XCAP, VROW = C::VROW, FR = C::FR;
    constexpr int VAL_ALIGN = (S % 4 == 0) ? 4 : 2;
    constexpr int NH = 4 + 2 * FB;                             // f32x2 accumulators per support
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* Wof_h = reinterpret_cast<__bf16*>(lds_raw);       // [s][o][32 f], 16-byte chunks XOR-swizzled by gml_wkey(o)
    __bf16* Wof_l = Wof_h + C::W_HALF;
    __bf16* W2_h = reinterpret_cast<__bf16*>(lds_raw + C::OFF_W2);   // [s][o][kq][hi 4 | lo 4] (features 32 + 4 kq + j), 64-byte rows

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);
    if (g0 >= g1) return;
    const bool loader = wave >= 8;

    if (GML_F4_DELAY & 1) GML_F4_IDLE();
    // ---- once per workgroup: W images, zeroed X areas (chunks at or beyond Fin are never written by a DMA and stay zero)
    // f16 pieces: the largest magnitude of every output column first (32 words at the start of the first X area: zeroed again below)
    uint32_t* cmax = reinterpret_cast<uint32_t*>(lds_raw + C::x_off(0));
    float winv_r[NOB];                                         // 1 / (scale of the lane's W columns)
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) winv_r[ob] = 1.f;
    if constexpr (F16) {
        if (tid < 32) cmax[tid] = 0u;
        __syncthreads();
        {
            const int o = tid & 31;
            constexpr int NF = FB ? 48 : 32;
            float m = 0.f;
            if (o < p.Fout)
                for (int i = tid >> 5; i < S * NF; i += C::NT / 32) {
                    const int s = i / NF, f = i % NF;
                    if (f < p.Fin) m = fmaxf(m, fabsf(p.w[(int64_t)(p.s0 + s) * p.w_ss + (int64_t)f * p.w_si + (int64_t)o * p.w_so]));
                }
            atomicMax(&cmax[o], __float_as_uint(m));
        }
        __syncthreads();
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) { float sc; gml_f16_scale_bits(cmax[ob * 16 + r16], sc, winv_r[ob]); }
    }
    for (int e = tid; e < S * 32 * 32; e += C::NT) {
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
    if constexpr (FB) {
        for (int e = tid; e < S * 32 * 16; e += C::NT) {
            const int f2 = e & 15, o = (e >> 4) & 31, s = e >> 9;
            const int f = 32 + f2;
            const float v = (f < p.Fin && o < p.Fout) ? p.w[(int64_t)(p.s0 + s) * p.w_ss + (int64_t)f * p.w_si + (int64_t)o * p.w_so] : 0.f;
            // [s][o][kq][hi 0..3 | lo 0..3]: the lane's K = 16 fragment pair is ONE 16-byte read
            const int i2 = ((s * 32 + o) * 4 + (f2 >> 2)) * 8 + (f2 & 3);
            if constexpr (F16) {
                float sc, inv;
                gml_f16_scale_bits(cmax[o], sc, inv);
                const float vs = v * sc;
                const _Float16 h = (_Float16)vs;
                reinterpret_cast<_Float16*>(W2_h)[i2] = h;
                reinterpret_cast<_Float16*>(W2_h)[i2 + 4] = (_Float16)(vs - (float)h);
            } else {
                const __bf16 h = (__bf16)v;
                const __bf16 l = (__bf16)(v - (float)h);
                W2_h[i2] = h;
                W2_h[i2 + 4] = l;
            }
        }
    }
    if constexpr (F16) __syncthreads();                        // (the column maxima sit in the first X area, zeroed next)
#pragma unroll
    for (int sl = 0; sl < (X1 ? 1 : 2); ++sl)
        for (int i = tid; i < C::X_BYTES / 16; i += C::NT)
            *reinterpret_cast<f32x4*>(lds_raw + C::x_off(sl) + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

#if GML_F4DBG & 4
    {   // debugging: every landing zone starts as NaN patterns (data read before it landed shows up as NaN rows)
        const float qn = __int_as_float(0x7fc00000);
        for (int i = tid; i < (int)(C::lds_bytes() - C::OFF_REC) / 4; i += C::NT) reinterpret_cast<float*>(lds_raw + C::OFF_REC)[i] = qn;
    }
#endif
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
        // a group's X window, whole blocks (the loaders with fewer value batches first)
        auto issue_x = [&](int g, const Geo& q) {
            if (!q.staged) return;
            const int nblk = (q.nwin + C::XRB - 1) / C::XRB;
            for (int b = NL - 1 - li; b < nblk; b += NL) {
                const int xo = (q.lo_a + C::XRB * b) * ldxb;
                const uint32_t xd = lds0 + C::x_off(g) + b * C::XBLK;
#pragma unroll
                for (int j = 0; j < C::XIB; ++j)
                    if (xon[j]) gml_dma16(rs_x, xd + j * 1024, xo + xsrc[j]);
            }
        };
        auto issue = [&](int it, int g, int c, const Geo& q, bool with_x) {
            const uint32_t ebuf = lds0 + C::OFF_EDGE + (it & 1) * C::EDGE_BYTES;
            if (q.staged) {
                int e0, n;
                chunk_of(q, c, e0, n);
                // ---- value rows
                if constexpr (!EP && S % 4 == 0) {
                    // CSR order, rows of whole 16-byte pieces: one contiguous byte range
                    const int npc = (n * VROW + 15) >> 4;
                    for (int j = li; 64 * j < npc; j += NL)
                        if (64 * j + lane < npc) gml_dma16(rs_val, ebuf + C::OFF_VAL + j * 1024, e0 * VROW + (64 * j + lane) * 16);
                } else if constexpr (!EP) {
                    // CSR order, 24-byte rows: two overlapping 16-byte lanes per row (bytes 0..15 and 8..23), 32 rows per instruction
                    const int nvi = (n + 31) >> 5;
                    for (int j = li; j < nvi; j += NL) {
                        const int e = 32 * j + (lane >> 1);
                        if (e < n) gml_dma16(rs_val, ebuf + C::OFF_VAL + j * 1024, (e0 + e) * VROW + (lane & 1) * 8);
                    }
                } else {
                    const int* epos_l = reinterpret_cast<const int*>(lds_raw + C::OFF_EPOS + (it & 1) * C::COL_BYTES);
                    constexpr int LPE = C::VROW_L / 16;                    // lanes (16 bytes each) per value row
                    constexpr int EPI = 64 / LPE;                          // value rows per instruction
                    constexpr int LSTEP = (S % 4 == 0) ? 16 : 8;           // source byte step between the lanes of a row
                    const int nvi = (n + EPI - 1) / EPI;
                    const int part = (lane % LPE) * LSTEP;
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
                            if (e < n) gml_dma16(rs_val, ebuf + C::OFF_VAL + (j0 + u) * 1024, voff[u]);
                        }
                    }
                }
                // ---- column ids: 256 per instruction
                const int nci = (n + 255) >> 8;
                for (int j = (li + 2) % NL; j < nci; j += NL)
                    if (256 * j + 4 * lane < n) gml_dma16(rs_col, ebuf + j * 1024, (e0 + 256 * j + 4 * lane) * 4);
                // ---- a new group: its X window
                if (c == 0 && with_x) issue_x(g, q);
            }
            if (c == 0 && li == NL - 1 && lane < 33) gml_dma16(rs_rp, lds0 + C::rp_off(g), (g * ROWS + 4 * lane) * 4);
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
        issue(0, g0, 0, q, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (GML_F4_DELAY & 16) GML_F4_IDLE();
        for (;;) {
            __builtin_amdgcn_s_barrier();                      // (B) item `it` complete in its buffers; every wave has left the other ones
            asm volatile("" ::: "memory");
            if (c == 0 && g + 3 < g1) dma_rec(g + 3);          // (entry (g - 1) & 3: group g - 1 is finished)
            int ng, nc; Geo nq;
            const bool has = next_of(g, c, q, ng, nc, nq);
            if (has) issue(it + 1, ng, nc, nq, !X1);
            if constexpr (X1) {
                if (has && nc == 0) {                          // the next item opens a group: its window goes where this group's is read
                    __builtin_amdgcn_s_barrier();              // (C) every compute wave has finished this group's aggregation
                    asm volatile("" ::: "memory");
                    issue_x(ng, nq);
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (GML_F4_DELAY & 16) GML_F4_IDLE();
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
        if (GML_F4_DELAY & 2) GML_F4_IDLE();

        // projection + stores of one group's tile: out tile = sum_s acc_s W_s (acc split on the fly = A fragments; K = 32: k = f =
        // 8 kq + j; K = 16: k = 4 kq + j <-> f = 32 + 4 kq + j)
        auto project = [&](f32x2 (&acc)[S][NH], int64_t r0, int nr, uint32_t out_rows) {
            // Order of the projection (pinned with sched_barrier, found the hard way -- DESIGN s4.1c): an MFMA that waits in the matrix
            // pipe behind its predecessors reads its operands when it STARTS, and an LDS load landing in one of those registers in
            // the meantime is not interlocked (hipcc assumes operands are read at issue and re-uses them for the next fragment
            // loads at once: wrong tiles on some waves of some launches).  So the loads for support s + 2 are issued only after the
            // VALU split of support s + 1, which cannot overtake the MFMAs of support s: two fragment sets, two split sets.
            float oscale[NOB];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) oscale[ob] = 1.f;
            auto mm = [&](f32x4 (&oacc)[NOB]) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) oacc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
            FT wh[2][NOB], wl[2][NOB];
            FT4 vh[2][NOB], vl[2][NOB];
            FT ah[2], al[2];
            FT4 bh[2], bl[2];
            float asc = 1.f;
            if constexpr (F16) {                               // the tile's scale (gml_spectconv_fwd3_impl.h), before the first split
                float m = 0.f;
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int h = 0; h < NH; ++h) m = fmaxf(fmaxf(fabsf(acc[s][h].x), fabsf(acc[s][h].y)), m);
                float ainv;
                gml_f16_scale_bits(gml_wave_max_bits(__float_as_uint(m)), asc, ainv);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oscale[ob] = ainv * winv_r[ob];
            }
            // `fa` (zero) is added to every fragment address and passes through a VALU statement behind each support's MFMAs: the
            // next loads then carry an address dependency on an instruction that issues IN ORDER behind those MFMAs
            int fa = 0;
            asm volatile("v_mov_b32 %0, 0" : "=v"(fa));
            auto frag = [&](int s, int st) {
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) {
                    const int o = ob * 16 + r16;               // B[k = f][n = o]: 8 consecutive f of column o
                    const int off = (s * 32 + o) * 32 + (((kq ^ gml_wkey(o)) & 3) << 3) + fa;
                    wh[st][ob] = *reinterpret_cast<const FT*>(Wof_h + off);
                    wl[st][ob] = *reinterpret_cast<const FT*>(Wof_l + off);
                    if constexpr (FB) {
                        const FT t = *reinterpret_cast<const FT*>(W2_h + ((s * 32 + o) * 4 + kq) * 8 + fa);
                        vh[st][ob] = FT4{t[0], t[1], t[2], t[3]};
                        vl[st][ob] = FT4{t[4], t[5], t[6], t[7]};
                    }
                }
            };
            auto split = [&](int s, int t) {
                const float av[8] = {acc[s][0].x, acc[s][0].y, acc[s][1].x, acc[s][1].y, acc[s][2].x, acc[s][2].y, acc[s][3].x, acc[s][3].y};
                if constexpr (F16) {
                    gml_split8_f16(av, asc, ah[t], al[t]);
                    if constexpr (FB) {
                        const f32x2 v0 = acc[s][4] * asc, v1 = acc[s][5] * asc;
                        const f16x2 h0 = __builtin_convertvector(v0, f16x2), h1 = __builtin_convertvector(v1, f16x2);
                        const f16x2 l0 = __builtin_convertvector(v0 - __builtin_convertvector(h0, f32x2), f16x2);
                        const f16x2 l1 = __builtin_convertvector(v1 - __builtin_convertvector(h1, f32x2), f16x2);
                        bh[t] = FT4{h0[0], h0[1], h1[0], h1[1]};
                        bl[t] = FT4{l0[0], l0[1], l1[0], l1[1]};
                    }
                } else {
                    gml_split8(av, ah[t], al[t]);
                    if constexpr (FB) {
                        bf16x2 h0, l0, h1, l1;
                        gml_split2(acc[s][4].x, acc[s][4].y, h0, l0);
                        gml_split2(acc[s][5].x, acc[s][5].y, h1, l1);
                        bh[t] = FT4{h0[0], h0[1], h1[0], h1[1]};
                        bl[t] = FT4{l0[0], l0[1], l1[0], l1[1]};
                    }
                }
            };
            frag(0, 0);
            if constexpr (S > 1) frag(1, 1);
            split(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int st = s & 1;
#ifdef GML_F4_ASM_MFMA
                // experiment: in-place accumulation (C = D) through inline asm, pads by hand
                asm volatile("s_nop 1");
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(oacc[ob]) : "v"(al[st]), "v"(wh[st][ob]));
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(oacc[ob]) : "v"(ah[st]), "v"(wl[st][ob]));
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(oacc[ob]) : "v"(ah[st]), "v"(wh[st][ob]));
                if constexpr (FB) {
                    asm volatile("s_nop 7");
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(oacc[ob]) : "v"(bl[st]), "v"(vh[st][ob]));
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(oacc[ob]) : "v"(bh[st]), "v"(vl[st][ob]));
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(oacc[ob]) : "v"(bh[st]), "v"(vh[st][ob]));
                    asm volatile("s_nop 7");
                }
                if (s == S - 1) asm volatile("s_nop 15\n\ts_nop 15");
#else
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oacc[ob] = gml_mfma_piece(al[st], wh[st][ob], oacc[ob]);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oacc[ob] = gml_mfma_piece(ah[st], wl[st][ob], oacc[ob]);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) oacc[ob] = gml_mfma_piece(ah[st], wh[st][ob], oacc[ob]);
                if constexpr (FB) {
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) oacc[ob] = gml_mfma_piece4(bl[st], vh[st][ob], oacc[ob]);
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) oacc[ob] = gml_mfma_piece4(bh[st], vl[st][ob], oacc[ob]);
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) oacc[ob] = gml_mfma_piece4(bh[st], vh[st][ob], oacc[ob]);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < S) split(s + 1, st ^ 1);           // VALU: issued behind the MFMAs above, in order
                __builtin_amdgcn_sched_barrier(0);
                if (s + 2 < S) {
                    asm volatile("v_mov_b32 %0, %0" : "+v"(fa));   // (in order behind the MFMAs above)
                    frag(s + 2, st);                           // the set the MFMAs above read
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            };
            f32x4 oacc[NOB];
            mm(oacc);
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
                        const float v = F16 ? fmaf(oacc[ob][reg], oscale[ob], bias_r[ob]) : oacc[ob][reg] + bias_r[ob];
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
                        float v = F16 ? fmaf(oacc[ob][reg], oscale[ob], bias_r[ob]) : oacc[ob][reg] + bias_r[ob];
                        if (relu) v = fmaxf(v, 0.f);
                        const int off = (o < p.Fout && lr < nr) ? (lr * (int)p.ldo + o) * 4 : 0x7fffff00;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, off, 0, 0);
                    }
                }
            }
        };

#if GML_F4DBG & 64
        auto check_w = [&](int slot) {                        // debugging: the W images against global memory
            for (int e = tid; e < S * 32 * 32; e += 512) {
                const int f = e & 31, o = (e >> 5) & 31, s = e >> 10;
                const float v = (f < p.Fin && o < p.Fout) ? p.w[(int64_t)(p.s0 + s) * p.w_ss + (int64_t)f * p.w_si + (int64_t)o * p.w_so] : 0.f;
                const __bf16 h = (__bf16)v;
                const __bf16 l = (__bf16)(v - (float)h);
                const int iof = (s * 32 + o) * 32 + ((((f >> 3) ^ gml_wkey(o)) & 3) << 3) + (f & 7);
                if ((float)Wof_h[iof] != (float)h || (float)Wof_l[iof] != (float)l) atomicAdd(&p.prof[slot], 1ull);
            }
            if constexpr (FB) {
                for (int e = tid; e < S * 32 * 16; e += 512) {
                    const int f2 = e & 15, o = (e >> 4) & 31, s = e >> 9;
                    const int f = 32 + f2;
                    const float v = (f < p.Fin && o < p.Fout) ? p.w[(int64_t)(p.s0 + s) * p.w_ss + (int64_t)f * p.w_si + (int64_t)o * p.w_so] : 0.f;
                    const __bf16 h = (__bf16)v;
                    const __bf16 l = (__bf16)(v - (float)h);
                    const int i2 = ((s * 32 + o) * 4 + (f2 >> 2)) * 8 + (f2 & 3);
                    if ((float)W2_h[i2] != (float)h || (float)W2_h[i2 + 4] != (float)l) atomicAdd(&p.prof[slot + 1], 1ull);
                }
            }
        };
        check_w(0);
#endif
        // One barrier per work item, ONE barrier site: a flat loop over the items (group, chunk) like fwd3's loop over groups; a new
        // group's state is set up where its first chunk starts, the projection follows its last chunk.
        int it = 0, g = g0, c = 0;
        Geo q = {};
        int row = 0, nr = 0, kbeg = 0, kend = 0, xoff = 0;
        uint32_t out_rows = 0;
        int64_t r0 = 0;
        bool rvalid = false;
        f32x2 acc[S][NH];
        for (;;) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // (B) item `it` has landed
            asm volatile("" ::: "memory");
            if (GML_F4_DELAY & 4) GML_F4_IDLE();
            if (c == 0) {
                q = geo_of(g);
                const int* rp_l = reinterpret_cast<const int*>(lds_raw + C::rp_off(g));
                const unsigned char* rec = lds_raw + C::OFF_REC + (g & 3) * 256;
                row = rec[16 + wave * 16 + r16];
                out_rows = reinterpret_cast<const uint32_t*>(rec + 16)[wave * 4 + kq];
                r0 = (int64_t)g * ROWS;
                nr = (int)min((int64_t)ROWS, p.nrows - r0);
                rvalid = row < nr;
                kbeg = rvalid ? rp_l[row] : 0;
                kend = rvalid ? rp_l[row + 1] : 0;
                // byte offset of (row cidx, features 8 kq ..) in the window: xoff + FR * 4 * cidx + 16 (cidx / XRB)
                xoff = C::x_off(g) + kq * 32 - q.lo_a * (FR * 4) - (q.lo_a / C::XRB) * 16;
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int h = 0; h < NH; ++h) acc[s][h] = f32x2{0.f, 0.f};
                if (!q.staged) {
                    // ---- column window wider than the staged rows: this kernel has no second road (an inlined global-gather road
                    //      set the register count of the whole kernel: 168 + spills).  The caller routes such batches elsewhere
                    //      (gml_spectconv_fwd_stage_window(); functional.fwd_groups); one that did not gets NaN rows, not wrong numbers.
                    const float qnan = __int_as_float(0x7fc00000);
                    if (rvalid)
                        for (int o = kq; o < p.Fout; o += 4) p.out[(r0 + row) * p.ldo + o] = qnan;
                    ++it;
                    if (++g >= g1) break;
                    if constexpr (X1) __builtin_amdgcn_s_barrier();    // (C) the loaders wait for it before the next group's window
                    continue;
                }
            }
            {
                // ---- aggregation (fp32 VALU, packed): acc[s][f] += val[k, s] * x[col[k], f] over the lane's edges inside this chunk
                const unsigned char* ebuf = lds_raw + C::OFF_EDGE + (it & 1) * C::EDGE_BYTES;
                const int* col_l = reinterpret_cast<const int*>(ebuf);
                const float* ea_l = reinterpret_cast<const float*>(ebuf + C::OFF_VAL);
                const int e0 = q.kb4 + c * q.cs;
                const int e1 = min(e0 + q.cs, q.kb4 + q.ne4);
#if GML_F4DBG & 8
                if constexpr (!EP) {                           // verify what landed against global memory
                    const int n = e1 - e0;
                    for (int i = tid; i < n * S; i += 512) {
                        const int kk_ = i / S, ss_ = i % S;
                        const float a = C::VROW_L == VROW ? ea_l[i] : ea_l[kk_ * 8 + (ss_ < 4 ? ss_ : ss_ + 2)], b = p.val[(int64_t)e0 * S + i];
                        if (__float_as_int(a) != __float_as_int(b)) atomicAdd(&p.prof[0], 1ull);
                        if (i == 0) atomicAdd(&p.prof[4], 1ull);
                    }
                    for (int i = tid; i < n; i += 512) {
                        const int a = col_l[i], b = (e0 + i < etot) ? p.col[e0 + i] : a;
                        if (a != b) atomicAdd(&p.prof[1], 1ull);
                    }
                    if (c == 0) {
                        const int* rp_l = reinterpret_cast<const int*>(lds_raw + C::rp_off(g));
                        const unsigned char* rec = lds_raw + C::OFF_REC + (g & 3) * 256;
                        for (int i = tid; i < q.nwin * FR; i += 512) {
                            const int rr = i / FR, f = i % FR;
                            const int cidx = q.lo_a + rr;
                            const float a = *reinterpret_cast<const float*>(lds_raw + (xoff - kq * 32) + cidx * (FR * 4) + ((cidx / C::XRB) << 4) + f * 4);
                            const float b = (cidx < p.nrows && f < p.Fin) ? p.x[(int64_t)cidx * p.ldx + f] : 0.f;
                            if (__float_as_int(a) != __float_as_int(b)) atomicAdd(&p.prof[2], 1ull);
                        }
                        for (int i = tid; i <= nr; i += 512) {
                            const int a = rp_l[i], b = p.rowptr[r0 + i];
                            if (a != b) atomicAdd(&p.prof[3], 1ull);
                        }
                        if (tid < 36) {
                            const int a = reinterpret_cast<const int*>(rec)[tid], b = p.ginfo[(int64_t)g * 36 + tid];
                            if (a != b) atomicAdd(&p.prof[5], 1ull);
                        }
                        {
                            const int rr = reinterpret_cast<const unsigned char*>(p.ginfo + (int64_t)g * 36 + 4)[wave * 16 + r16];
                            const uint32_t orr = reinterpret_cast<const uint32_t*>(p.ginfo + (int64_t)g * 36 + 4)[wave * 4 + kq];
                            if (rr != row || orr != out_rows) atomicAdd(&p.prof[6], 1ull);
                        }
                    }
                }
#endif
                int k = max(kbeg, e0) - e0;
                const int ke = min(kend, e1) - e0;
#ifdef GML_F4_UNIFORM
                // experiment: no lane leaves the edge loop before the whole wave is done (EXEC never changes inside it); a lane
                // past its last edge keeps walking clamped positions with its support values scaled by zero
                if (__builtin_amdgcn_ballot_w64(k < ke) != 0) {
                    const int nmax = e1 - e0 - 1;
                    struct Ops { float e[S]; f32x4 t0, t1, t2; };
                    auto fetchu = [&](Ops& o, int kk, int cidx) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(ea_l + kk * 8), b = *reinterpret_cast<const f32x4*>(ea_l + kk * 8 + 4);
                        o.e[0] = a.x; o.e[1] = a.y; o.e[2] = a.z; o.e[3] = a.w; o.e[4] = b.z; o.e[5] = b.w;
                        const int off = xoff + cidx * (FR * 4) + ((cidx / C::XRB) << 4);
                        o.t0 = *reinterpret_cast<const f32x4*>(lds_raw + off);
                        o.t1 = *reinterpret_cast<const f32x4*>(lds_raw + off + 16);
                        if constexpr (FB) o.t2 = *reinterpret_cast<const f32x4*>(lds_raw + off + 128 - 16 * kq);
                    };
                    auto fmau = [&](const Ops& o, float act) {
                        f32x2 xv[NH];
                        xv[0] = f32x2{o.t0.x, o.t0.y}; xv[1] = f32x2{o.t0.z, o.t0.w}; xv[2] = f32x2{o.t1.x, o.t1.y}; xv[3] = f32x2{o.t1.z, o.t1.w};
                        if constexpr (FB) { xv[4] = f32x2{o.t2.x, o.t2.y}; xv[5] = f32x2{o.t2.z, o.t2.w}; }
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const float ev = o.e[s] * act;
                            const f32x2 e2 = f32x2{ev, ev};
#pragma unroll
                            for (int h = 0; h < NH; ++h) acc[s][h] = e2 * xv[h] + acc[s][h];
                        }
                    };
                    auto cl = [&](int i) { return min(max(i, 0), nmax); };
                    Ops A, B;
                    fetchu(A, cl(k), col_l[cl(k)]);
                    int cn = col_l[cl(k + 1)];
                    for (;;) {
                        const int c2 = col_l[cl(k + 2)];
                        fetchu(B, cl(k + 1), cn);
                        fmau(A, (k >= 0 && k < ke) ? 1.f : 0.f);
                        ++k;
                        if (__builtin_amdgcn_ballot_w64(k < ke) == 0) break;
                        cn = col_l[cl(k + 2)];
                        fetchu(A, cl(k + 1), c2);
                        fmau(B, (k >= 0 && k < ke) ? 1.f : 0.f);
                        ++k;
                        if (__builtin_amdgcn_ballot_w64(k < ke) == 0) break;
                    }
                }
                if (false) {
#else
                if (k < ke) {
#endif
                    // software pipeline as in fwd3: operands of edge k + 1 and the column id of edge k + 2 are requested before
                    // the packed FMAs of edge k; two register sets, no rotation moves
                    struct Ops { float e[S]; f32x4 t0, t1, t2; };
                    auto fetch = [&](Ops& o, int kk, int cidx) {
                        if constexpr (C::VROW_L == VROW) gml_load_row<S, VAL_ALIGN>(ea_l + kk * S, o.e);
                        else {                                 // gathered 24-byte rows: [v0 v1 v2 v3 | v2 v3 v4 v5]
                            const f32x4 a = *reinterpret_cast<const f32x4*>(ea_l + kk * 8), b = *reinterpret_cast<const f32x4*>(ea_l + kk * 8 + 4);
                            o.e[0] = a.x; o.e[1] = a.y; o.e[2] = a.z; o.e[3] = a.w; o.e[4] = b.z; o.e[5] = b.w;
                        }
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
#ifdef GML_F4_SCALARFMA
#pragma unroll
                            for (int h = 0; h < NH; ++h) {
                                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[s][h].x) : "v"(o.e[s]), "v"(xv[h].x));
                                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[s][h].y) : "v"(o.e[s]), "v"(xv[h].y));
                            }
#else
                            // The support value of an ODD position of its row goes through a v_mov first.  Left alone, hipcc broadcasts it
                            // straight out of the odd register of the loaded pair -- v_pk_fma_f32 acc, x, v[a:a+1] op_sel:[0,1,0] (the LOW
                            // product reads the HIGH half of src1) -- and on gfx950 that form lost, about once in 10^8 issues, the low
                            // product on lanes 48..63 of one wave: a row's last edge missing from ONE (support, feature) aggregate, 3-8 of
                            // 768 fresh-process launches of the chunked road wrong (the mechanism is not known; found by dumping the
                            // aggregates of failing launches and matching the difference to single edge terms -- DESIGN s4.1c).  With the
                            // copy the compiler emits the op_sel_hi form every other kernel uses: 0 of 1536 (GML_F4_OPSEL restores the old
                            // code for the A/B).
                            float ev = o.e[s];
#ifndef GML_F4_OPSEL
                            if (s & 1) asm volatile("v_mov_b32 %0, %1" : "=v"(ev) : "v"(o.e[s]));
#endif
                            const f32x2 e2 = f32x2{ev, ev};
#pragma unroll
                            for (int h = 0; h < NH; ++h) acc[s][h] = e2 * xv[h] + acc[s][h];
#endif
                        }
#ifdef GML_F4_NOPS
                        asm volatile("s_nop 7\n\ts_nop 7");
#endif
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
            ++it;
            if (++c < q.nch) continue;
            c = 0;
            // The pipelined loop leaves its last prefetch (the clamped "next" edge) in flight at its exits: drain it before the
            // projection takes the registers over (the projection's own loads are ordered by hand, see `project`).
            __builtin_amdgcn_sched_barrier(0);                 // (nothing of the projection may be scheduled above the wait)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (X1) {
                if (g + 1 < g1) __builtin_amdgcn_s_barrier();  // (C) this group's window is free: the next one lands beside the projection
                __builtin_amdgcn_sched_barrier(0);
            }
            if (GML_F4_DELAY & 8) { GML_F4_IDLE(); __builtin_amdgcn_sched_barrier(0); }
#ifdef GML_F4_EXECCHK
            if (__builtin_amdgcn_read_exec() != ~0ull && p.prof) atomicAdd(&p.prof[7], 1ull);
#endif
#if GML_F4DBG & 256
            if (p.hout != nullptr && rvalid) {                 // debugging: the aggregate H[row][s][48] as the projection receives it
                float* hr = p.hout + ((r0 + row) * S) * 48;
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int h = 0; h < 4; ++h) { hr[s * 48 + 8 * kq + 2 * h] = acc[s][h].x; hr[s * 48 + 8 * kq + 2 * h + 1] = acc[s][h].y; }
                    if constexpr (FB) {
                        hr[s * 48 + 32 + 4 * kq] = acc[s][4].x; hr[s * 48 + 33 + 4 * kq] = acc[s][4].y;
                        hr[s * 48 + 34 + 4 * kq] = acc[s][5].x; hr[s * 48 + 35 + 4 * kq] = acc[s][5].y;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
            project(acc, r0, nr, out_rows);
            if (++g >= g1) break;
        }
#if GML_F4DBG & 64
        check_w(2);
#endif
    }
}

template <int S, int FB, int NOB>
int gml_launch_fwd4(const GmlFwdParams& p, dim3 grid, hipStream_t st);

#define GML_FWD4_LAUNCH_F(SV, FBV, NOBV, EPV, X1V, F16V)                                                     \
    {                                                                                                        \
        GML_ALLOW_BIG_LDS(rc_, (&gml_k_spectconv_fwd4<SV, FBV, NOBV, EPV, X1V, F16V>), 160 * 1024)           \
        if (rc_ != hipSuccess) return (int)rc_;                                                              \
        const size_t lds_ = GmlFwd4Cfg<SV, FBV, EPV, X1V>::lds_bytes();                                      \
        hipLaunchKernelGGL((gml_k_spectconv_fwd4<SV, FBV, NOBV, EPV, X1V, F16V>), grid, dim3(GML_FWD4_NT), lds_, st, p); \
        return gml_launch_status();                                                                          \
    }
#define GML_FWD4_LAUNCH(SV, FBV, NOBV, EPV, X1V)                                                             \
    {                                                                                                        \
        if (p.flags & GML_F16X3) GML_FWD4_LAUNCH_F(SV, FBV, NOBV, EPV, X1V, true)                            \
        GML_FWD4_LAUNCH_F(SV, FBV, NOBV, EPV, X1V, false)                                                    \
    }
// GML_FWD_ONEWIN (48-feature shapes only): the single-window form, for batches whose groups need edge chunks
#define GML_DEFINE_FWD4(SV, FBV, NOBV)                                                                       \
    template <>                                                                                              \
    int gml_launch_fwd4<SV, FBV, NOBV>(const GmlFwdParams& p, dim3 grid, hipStream_t st) {                   \
        if constexpr (FBV == 1) {                                                                            \
            if (p.flags & GML_FWD_ONEWIN) {                                                                  \
                if (p.epos != nullptr) GML_FWD4_LAUNCH(SV, FBV, NOBV, true, true)                            \
                GML_FWD4_LAUNCH(SV, FBV, NOBV, false, true)                                                  \
            }                                                                                                \
        }                                                                                                    \
        if (p.epos != nullptr) GML_FWD4_LAUNCH(SV, FBV, NOBV, true, false)                                   \
        GML_FWD4_LAUNCH(SV, FBV, NOBV, false, false)                                                         \
    }
