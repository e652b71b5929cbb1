// Fused forward, 128-row / 8-wave geometry with an LDS-DMA landing ring (bf16x3 projection; Fin <= 32, Fout <= 32, S in {4, 8},
// float4-addressable x): same function as gml_k_spectconv_fwd2 (gml_spectconv_fwd2_impl.h)
//
//   out[r, :] = act( sum_s (sum_{k in row r} val[pos(k), s] x[col[k], :]) W_s + b )      libs/spect_conv.py:76-80, :93-94
//
// What changed against fwd2: a group's CSR slice, value rows, X window and record no longer travel
// HBM -> registers -> ds_write -> barrier.  Every wave issues its share of `buffer_load_dwordx4 ... lds` instructions
// (1 KiB per wave-instruction, lane l lands at base + 16 l, no VGPR, no ds_write) that fill the OTHER of two LDS slots while
// the current slot is aggregated and projected; the only per-group synchronisation is one `s_waitcnt vmcnt(stores)` +
// `s_barrier`.  The records run three groups ahead in a 4-entry ring, so no dependent global round trip is left on the
// per-group path.  Layouts are what a lane-linear landing zone allows:
//   * col / value rows / row pointers: verbatim copies starting at the 16-byte aligned edge kb & ~3 (the aggregation
//     indexes them with k - (kb & ~3); column ids stay absolute);
//   * X window: rows lo8 .. (lo8 = window start & ~7), 8 rows (1 KiB) per instruction, blocks 1040 bytes apart: the 16-byte
//     pad per block spreads the b128 gathers of random rows over all 16 bank groups ((block + 8 (row & 1) + chunk) mod 16),
//     which the padded 144-byte rows of fwd2 did per row; address = base + 128 row + 2 (row & ~7), three VALU ops;
//   * learned supports (EP): value rows are gathered through epos by the DMA itself (per-lane source address
//     val + 32 epos[k] + 16 (lane & 1)); the positions of group g + 2 land in a wave-private chunk of ONE buffer while the
//     wave reads the positions of g + 1 from it (in-order inside the wave, no second buffer);
//   * out-of-range protection is the buffer descriptors' (rows past the end, edges past the end read zeros), lanes past a
//     group's extent are switched off by EXEC: no clamped duplicate loads, no predicates on VGPR results.
// The Hadamard branch reads its own x rows from the window (the window is widened to include the group's rows).
// Groups outside the LDS capacities take the global-gather path of fwd2 (same results).
#pragma once
#include "gml_common.h"
#include "gml_spectconv_impl.h"

#ifndef GML_FWABL
#define GML_FWABL 0
#endif
// cache policy of the wide output stores (0 = default, 2 = nt; A/B: tools/build_variant.py f3nt -DGML_FWD3_ST_AUX=2)
#ifndef GML_FWD3_ST_AUX
#define GML_FWD3_ST_AUX 0
#endif
#ifdef GML_FWD2_TIMING
#define GML_TF3(i) do { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); tacc_[i] += t_ - tprev_; tprev_ = t_; } while (0)
#else
#define GML_TF3(i)
#endif

typedef __attribute__((address_space(3))) void gml_lds_void;

// Buffer descriptor of a plain array (raw buffer, byte offsets, hardware range check: bytes at or beyond nbytes read 0).
// Built by hand as four dwords so that it can be an "s" operand of the DMA statement below.
__device__ __forceinline__ u32x4 gml_raw_rsrc(const void* base, uint32_t nbytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, nbytes, 0x00020000u};
}

// One LDS-DMA wave-instruction: 16 bytes per active lane from rs[voff] to LDS byte address lds_addr + 16 * lane (lds_addr
// wave-uniform).  Inline asm, not __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler orders every later ds_read that
// it cannot prove disjoint from the landing zone behind `s_waitcnt vmcnt(0)`, which drains the ring it is meant to keep
// in flight (seen in the ISA of the first version of this kernel); an asm statement is outside its bookkeeping, the
// kernel counts its DMAs itself.  M0 (the landing address) is compiler-reserved: saved and restored inside the statement;
// `s_nop 2`: with the two s_mov before it, the five wait states between a VALU write of an SGPR (a spilled descriptor
// dword restored by v_readlane right before the statement -- invisible to the hazard recogniser) and the VMEM that reads it;
// it also covers the one wait state between the M0 write and the LDS-DMA.
__device__ __forceinline__ void gml_dma16(u32x4 rs, uint32_t lds_addr, int voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}

// 4 x 4 transpose between the registers of a lane and the lanes of its quad: out[r] of quad lane b = in[b] of quad lane r
// (two butterfly stages of DPP quad_perm moves, 16 VALU operations)
__device__ __forceinline__ void gml_quad_transpose(f32x4& v, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    auto xch = [](float send, int ctrl_is_b1) {
        const int x = __float_as_int(send);
        return __int_as_float(ctrl_is_b1 ? __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true)      // quad_perm [2,3,0,1]
                                         : __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    };
    {   // stage 1: 2 x 2 blocks, partner lane ^ 1, register pairs (0,1) and (2,3)
        const float y0 = xch(b0 ? v[0] : v[1], 0), y1 = xch(b0 ? v[2] : v[3], 0);
        if (b0) { v[0] = y0; v[2] = y1; } else { v[1] = y0; v[3] = y1; }
    }
    {   // stage 2: partner lane ^ 2, register pairs (0,2) and (1,3)
        const float y0 = xch(b1 ? v[0] : v[2], 1), y1 = xch(b1 ? v[1] : v[3], 1);
        if (b1) { v[0] = y0; v[1] = y1; } else { v[2] = y0; v[3] = y1; }
    }
}

template <int S>
struct GmlFwd3Cfg {
    static constexpr int ROWS = 128;
    static constexpr int NLOAD = 4;                            // loader waves (one wave's LDS-DMA stream tops out near 1 KiB per ~100 cycles)
    static constexpr int NT = 512 + 64 * NLOAD;                // 8 compute waves + the loader waves: 3 waves per SIMD
    static constexpr int NTC = 512;
    static constexpr int XCAP = 200;                           // staged window rows incl. the <= 7 rows of alignment slack
    static constexpr int XBLK = 1040;                          // bytes between 8-row blocks of the window
    static constexpr int W_HALF = S * 32 * 32;                 // bf16 elements of one (hi or lo) W image [s][o][f]
    static constexpr int W_BYTES = 4 * W_HALF;
    static constexpr int REC_BYTES = 4 * 256;                  // ring of 4 group records (144 of 256 bytes used)
    static constexpr int RP_BYTES = 528;                       // 132 row pointers
    static constexpr int X_BYTES = XCAP / 8 * XBLK;
    static constexpr int AVAIL = 160 * 1024 - W_BYTES - REC_BYTES - 2 * (RP_BYTES + X_BYTES);
    static constexpr int ECAP_RAW = AVAIL / (2 * (4 + 4 * S) + 8);
    static constexpr int ECAP = ECAP_RAW >= 1024 ? 1024 : ECAP_RAW / 64 * 64;      // staged edges per group (S = 8: 960)
    static constexpr int COL_BYTES = ECAP * 4, VAL_BYTES = ECAP * S * 4;
    static constexpr int SLOT_BYTES = RP_BYTES + COL_BYTES + VAL_BYTES + X_BYTES;
    static constexpr int OFF_REC = W_BYTES, OFF_EPOS = OFF_REC + REC_BYTES, OFF_SLOT = OFF_EPOS + 2 * COL_BYTES;   // two position buffers
    static constexpr int OFF_COL = RP_BYTES, OFF_VAL = OFF_COL + COL_BYTES, OFF_X = OFF_VAL + VAL_BYTES;   // inside a slot
    static constexpr size_t lds_bytes() { return (size_t)OFF_SLOT + 2 * (size_t)SLOT_BYTES; }
    static_assert(SLOT_BYTES % 16 == 0 && OFF_SLOT % 16 == 0, "16-byte aligned landing zones");
};

// MIX: the ML3Layer Hadamard branch (F2 <= 8) of the group's own rows rides along (see fwd2); EP: value rows through p.epos
// EPL: epilogue (GmlFwdParams::epl): 0 sum over supports, 1 ConCat column blocks, 2 depthwise scale-then-one-projection
// F16: the projection (and the Hadamard branch) on f16 (hi, lo) pieces -- GML_F16X3: the aggregates of a 16-row tile share one
// power-of-two scale (the tile's largest magnitude just below 2^15), every output column of W its own; undone in the epilogue
template <int S, int NOB, bool MIX, bool EP, int EPL = 0, bool F16 = false>
__global__ __launch_bounds__(GmlFwd3Cfg<S>::NT, 1) void gml_k_spectconv_fwd3(const GmlFwdParams p) {
    using C = GmlFwd3Cfg<S>;
    using FT = typename GmlPiece<F16>::T;
    static_assert(S % 4 == 0, "float4 value rows");
    static_assert(!F16 || EPL == 0, "f16 pieces: the sum-over-supports epilogue");
    static_assert(EPL == 0 || (!MIX && !EP), "epilogue variants: plain SpectConv calls");
    constexpr bool OWNX = MIX || EPL == 2;                     // the lane's own x row is needed (Hadamard branch / depthwise self term)
    constexpr int SW = EPL == 2 ? 1 : S;                       // W images (depthwise: one matrix)
    constexpr int ROWS = C::ROWS, ECAP = C::ECAP, XCAP = C::XCAP;
    constexpr int VROW = S * 4;                                // bytes of a value row
    constexpr int LPE = S / 4;                                 // lanes (16 bytes each) per value row
    constexpr int EPI = 64 / LPE;                              // value rows per DMA instruction
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* Wof_h = reinterpret_cast<__bf16*>(lds_raw);       // [s][o][f], 16-byte chunks XOR-swizzled by gml_wkey(o)
    __bf16* Wof_l = Wof_h + C::W_HALF;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);
    if (g0 >= g1) return;
    const bool loader = wave >= 8;

    // ---- once per workgroup: W image, zeroed X areas (chunks at or beyond Fin are never written by a DMA and stay zero)
    // f16 pieces: first the largest magnitude of every output column (32 words of slot 0's column area, not yet in use)
    uint32_t* cmax = reinterpret_cast<uint32_t*>(lds_raw + C::OFF_SLOT + C::OFF_COL);
    if constexpr (F16) {
        if (tid < 32) cmax[tid] = 0u;
        __syncthreads();
        {   // thread = (column o = tid & 31, slice tid >> 5 of the (s, f) pairs): one atomic per thread, coalesced over o
            const int o = tid & 31;
            float m = 0.f;
            if (o < p.Fout)
                for (int i = tid >> 5; i < SW * 32; i += C::NT / 32) {
                    const int s = i >> 5, f = i & 31;
                    if (f < p.Fin) m = fmaxf(m, fabsf(p.w[(int64_t)(p.s0 + s) * p.w_ss + (int64_t)f * p.w_si + (int64_t)o * p.w_so]));
                }
            atomicMax(&cmax[o], __float_as_uint(m));
        }
        __syncthreads();
    }
    for (int e = tid; e < SW * 32 * 32; e += C::NT) {
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
    // depthwise: [S + 1][32] scales right behind the single hi W image (bytes [0, 2048)); the lo image keeps its place at byte
    // W_HALF * 2 = S * 2048 (>= 8192), so [2048, 2048 + 128 (S + 1)) is free for S in {4, 8} (r03: it sat at byte 8192 -- ON the
    // lo image when S = 4; found by the randomised sweep, tools/fuzz_parity.py)
    float* ds_l = reinterpret_cast<float*>(lds_raw + 2048);
    static_assert(EPL != 2 || (2048 + 128 * (S + 1) <= S * 2048), "depthwise scales overlap the lo W image");
    if constexpr (EPL == 2) {
        for (int e = tid; e < (S + 1) * 32; e += C::NT) {
            const int f = e & 31, s = e >> 5;
            ds_l[e] = (f < p.Fin && (s < S || p.ds_self)) ? p.ds[s * p.Fin + f] : 0.f;
        }
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
        for (int i = tid; i < C::X_BYTES / 16; i += C::NT)
            *reinterpret_cast<f32x4*>(lds_raw + C::OFF_SLOT + sl * C::SLOT_BYTES + C::OFF_X + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // a group's geometry from its record in the ring (wave-uniform)
    struct Geo { int kb4, ne4, lo8, nwin8; bool staged; };
    auto geo_of = [&](int g, bool bigv) -> Geo {
        const int4 v = *reinterpret_cast<const int4*>(lds_raw + C::OFF_REC + (g & 3) * 256);
        const int kb = __builtin_amdgcn_readfirstlane(v.x), ne = __builtin_amdgcn_readfirstlane(v.y);
        const int lo = __builtin_amdgcn_readfirstlane(v.z), nwin = __builtin_amdgcn_readfirstlane(v.w);
        const int r0 = g * ROWS;
        const int nr = (int)min((int64_t)ROWS, p.nrows - (int64_t)r0);
        int wlo = ne > 0 ? lo : r0, whi = ne > 0 ? lo + nwin : r0;
        if constexpr (OWNX) { wlo = min(wlo, r0); whi = max(whi, r0 + nr); }     // the group's own rows (Hadamard branch, depthwise self term)
        Geo q;
        q.kb4 = kb & ~3; q.ne4 = ne + (kb & 3);
        q.lo8 = wlo & ~7; q.nwin8 = whi - q.lo8;
        q.staged = q.ne4 <= ECAP && q.nwin8 <= XCAP && !bigv;
        return q;
    };
    // value rows beyond 32-bit byte offsets: every group takes the global-gather path (all waves must agree)
    const int etot = p.rowptr[p.nrows];
    const bool bigv = (uint64_t)etot * VROW > 0xffffff00ull;

#ifdef GML_FWD2_TIMING
    unsigned tacc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tprev_ = 0;
#endif

    if (loader) {
        // =====================================================================================================
        // Loader wave: all LDS-DMA of the workgroup.  Trip g: (barrier) -> data of group g + 1 into the other slot, value
        // positions of g + 2, record of g + 3 -> wait until everything has landed -> (next barrier).  The compute waves
        // never issue or wait for a memory instruction except their output stores.
        // =====================================================================================================
        const uint32_t lds0 = (uint32_t)(uintptr_t)((gml_lds_void*)lds_raw);   // LDS byte address of the dynamic segment
        // buffer descriptors: the hardware range check replaces every index clamp (out-of-range lanes read zeros)
        const u32x4 rs_rec = gml_raw_rsrc(p.ginfo, (uint32_t)p.ngroups * (GML_GREC_INTS(128) * 4));
        const u32x4 rs_rp = gml_raw_rsrc(p.rowptr, (uint32_t)(p.nrows + 1) * 4u);
        const u32x4 rs_col = gml_raw_rsrc(p.col, bigv ? 0u : (uint32_t)etot * 4u);
        const u32x4 rs_val = gml_raw_rsrc(p.val, bigv ? 0u : (uint32_t)etot * VROW);
        const u32x4 rs_epos = gml_raw_rsrc(p.epos, (EP && !bigv) ? (uint32_t)etot * 4u : 0u);
        const u32x4 rs_x = gml_raw_rsrc(p.x, (uint32_t)(p.nrows * p.ldx) * 4u);
        const int ldxb = (int)p.ldx * 4;
        constexpr int NL = C::NLOAD;
        const int li = wave - 8;                               // the DMA instructions of a group are dealt round robin to the loaders
        auto dma_rec = [&](int g) {                            // record of group g -> ring entry g & 3
            if (li == NL - 1 && lane < 9) gml_dma16(rs_rec, lds0 + C::OFF_REC + (g & 3) * 256, g * (GML_GREC_INTS(128) * 4) + lane * 16);
        };
        auto dma_epos = [&](int g, const Geo& q) {             // group g's value positions (256 per instruction) -> buffer g & 1
            const int nci = (q.ne4 + 255) >> 8;
            for (int j = li; j < nci; j += NL)
                if (256 * j + 4 * lane < q.ne4) gml_dma16(rs_epos, lds0 + C::OFF_EPOS + (g & 1) * C::COL_BYTES + j * 1024, (q.kb4 + 256 * j + 4 * lane) * 4);
        };
        // data of group g -> slot g & 1 (needs, with EP, the positions of g in the epos buffer); then positions of g + 1, record of g + 2
        auto issue = [&](int g, const Geo& q) {
            const uint32_t slot = lds0 + C::OFF_SLOT + (g & 1) * C::SLOT_BYTES;
            if (q.staged && !(GML_FWABL & 16)) {
                // value rows: four instructions per batch (with EP: their four position reads first, one wait)
                const int nvi = (q.ne4 + EPI - 1) / EPI;
                const int part = (lane % LPE) * 16;
                const int* epos_l = reinterpret_cast<const int*>(lds_raw + C::OFF_EPOS + (g & 1) * C::COL_BYTES);
                for (int j0 = 4 * li; j0 < nvi; j0 += 4 * NL) {
                    int voff[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = min((j0 + u) * EPI + lane / LPE, ECAP - 1);
                        if constexpr (EP) voff[u] = epos_l[e] * VROW + part;
                        else voff[u] = (q.kb4 + e) * VROW + part;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = (j0 + u) * EPI + lane / LPE;
                        if (e < q.ne4) gml_dma16(rs_val, slot + C::OFF_VAL + (j0 + u) * 1024, voff[u]);
                    }
                }
                // X window: 8 rows per instruction
                const int nxi = (q.nwin8 + 7) >> 3;
                const int c4 = (lane & 7) * 4;
                const int i0 = (NL - 1 - li);                  // (the loaders with fewer value batches first)
                int xo = (q.lo8 + 8 * i0 + (lane >> 3)) * ldxb + c4 * 4;
                uint32_t xd = slot + C::OFF_X + i0 * C::XBLK;
                for (int i = i0; i < nxi; i += NL) {
                    if (c4 < p.Fin) gml_dma16(rs_x, xd, xo);
                    xo += 8 * NL * ldxb; xd += NL * C::XBLK;
                }
                // column ids: 256 per instruction
                const int nci = (q.ne4 + 255) >> 8;
                for (int j = (li + 2) % NL; j < nci; j += NL)
                    if (256 * j + 4 * lane < q.ne4) gml_dma16(rs_col, slot + C::OFF_COL + j * 1024, (q.kb4 + 256 * j + 4 * lane) * 4);
            }
            if (li == NL - 1 && lane < 33) gml_dma16(rs_rp, slot, (g * ROWS + 4 * lane) * 4);
            if constexpr (EP) {
                if (g + 1 < g1) { const Geo q1 = geo_of(g + 1, bigv); if (q1.staged) dma_epos(g + 1, q1); }
            }
            if (g + 2 < g1) dma_rec(g + 2);
        };

        dma_rec(g0);
        if (g0 + 1 < g1) dma_rec(g0 + 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // (A) records landed, W image and zeroed X areas complete
        {
            const Geo q = geo_of(g0, bigv);
            if constexpr (EP) {
                if (q.staged) dma_epos(g0, q);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                  // (A2) every loader's share of the first positions has landed
            }
            issue(g0, q);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GML_FWD2_TIMING
        tprev_ = (unsigned)__builtin_readcyclecounter();
#endif
        for (int g = g0; g < g1; ++g) {
            __builtin_amdgcn_s_barrier();                      // (B) slot g & 1 complete; every wave has left slot (g + 1) & 1
            asm volatile("" ::: "memory");
            GML_TF3(5);
            if (g + 1 < g1) {
                const Geo qn = geo_of((GML_FWABL & 8) ? g0 : g + 1, bigv);
                issue(g + 1, qn);
            }
            GML_TF3(2);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            GML_TF3(0);
        }
    } else {
        // =====================================================================================================
        // Compute waves: one 16-row tile each
        // =====================================================================================================
        float bias_r[NOB];
        float winv_r[NOB];                                     // f16 pieces: 1 / (scale of the lane's W column)
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            bias_r[ob] = (p.bias && ob * 16 + r16 < p.Fout) ? p.bias[ob * 16 + r16] : 0.f;
            winv_r[ob] = 1.f;
            if constexpr (F16) { float sc; gml_f16_scale_bits(cmax[ob * 16 + r16], sc, winv_r[ob]); }
        }
        // Wide output stores (one 16-byte store per lane and 16-column block instead of four 4-byte ones, the Hadamard columns
        // merged into the conv tile they complete): possible when the row is written in whole float4 chunks -- columns
        // [0, Fout (+ F2)) with mix_col = Fout, a multiple of 4 in total, float4-addressable rows.  ZINC: 30 + 2 = 32.
        const int ncols = p.Fout + (MIX ? p.F2 : 0);
        const int m0 = MIX ? (p.mix_col & 15) : 0;
        const bool wide = !(GML_FWABL & 32) && ncols % 4 == 0 && p.ldo % 4 == 0 && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0) &&
                          (!MIX || (p.mix_col == p.Fout && (p.mix_col >> 4) == NOB - 1 && m0 + p.F2 <= 16 && p.F2 <= 8));
        // B[k = f][n = c] of the Hadamard product.  Narrow stores: c < F2 -> w11 row c, F2 <= c < 2 F2 -> w12 row c - F2 (partner
        // column c + F2).  Wide stores: w11 row i at column m0 + i -- where the product belongs in the conv tile -- and
        // w12 row i at column (m0 + i) ^ 8 (partner = lane ^ 8).
        FT mwh, mwl;
        float mbias = 0.f, mwinv = 1.f;
        if constexpr (MIX) {
            int r11 = -1, r12 = -1;                            // the w11 / w12 row this lane's column carries
            if (wide) {
                if (r16 >= m0 && r16 < m0 + p.F2) r11 = r16 - m0;
                else if ((r16 ^ 8) >= m0 && (r16 ^ 8) < m0 + p.F2) r12 = (r16 ^ 8) - m0;
            } else {
                if (r16 < p.F2) r11 = r16;
                else if (r16 < 2 * p.F2) r12 = r16 - p.F2;
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f = 8 * kq + j;
                v[j] = (f < p.Fin && r11 >= 0) ? p.w11[r11 * p.Fin + f] : ((f < p.Fin && r12 >= 0) ? p.w12[r12 * p.Fin + f] : 0.f);
            }
            if constexpr (F16) {                               // the column's scale: maximum over the four k groups of the lane's column
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
            if (r11 >= 0) mbias = p.b11 ? p.b11[r11] : 0.f;
            else if (r12 >= 0) mbias = p.b12 ? p.b12[r12] : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // (A)
        if constexpr (EP) __builtin_amdgcn_s_barrier();        // (A2)
#ifdef GML_FWD2_TIMING
        tprev_ = (unsigned)__builtin_readcyclecounter();
#endif
        for (int g = g0; g < g1; ++g) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // (B)
            asm volatile("" ::: "memory");
            GML_TF3(1);
            const Geo q = geo_of(g, bigv);
            const unsigned char* slot = lds_raw + C::OFF_SLOT + (g & 1) * C::SLOT_BYTES;
            const int* rp_l = reinterpret_cast<const int*>(slot);
            const int* col_l = reinterpret_cast<const int*>(slot + C::OFF_COL);
            const float* ea_l = reinterpret_cast<const float*>(slot + C::OFF_VAL);
            const unsigned char* rec = lds_raw + C::OFF_REC + (g & 3) * 256;
            const int row = rec[16 + wave * 16 + r16];
            const uint32_t out_rows = reinterpret_cast<const uint32_t*>(rec + 16)[wave * 4 + kq];
            const int64_t r0 = (int64_t)g * ROWS;
            const int nr = (int)min((int64_t)ROWS, p.nrows - r0);
            const bool rvalid = row < nr;
            // byte offset of (row c, features 8 kq ..) in the window: xoff + 128 c + 2 (c & ~7)
            const int xoff = C::OFF_SLOT + (g & 1) * C::SLOT_BYTES + C::OFF_X + kq * 32 - q.lo8 * 130;
            const int kbeg = rvalid ? rp_l[row] : 0;
            const int kend = rvalid ? rp_l[row + 1] : 0;
            float xrow[OWNX ? 8 : 1];                          // the lane's own x row, features 8*kq..8*kq+7
            GML_TF3(3);

            // ---- aggregation (fp32 VALU, packed): acc[s][f] += val[k, s] * x[col[k], f], f = 8*kq .. 8*kq+7
            f32x2 acc[S][4];
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int h = 0; h < 4; ++h) acc[s][h] = f32x2{0.f, 0.f};
            if (q.staged) {
                // Software pipeline over the lane's edges: the operands of edge k + 1 (value row, two x chunks) and the
                // column id of edge k + 2 are requested before the 32 packed FMAs of edge k, so the two dependent LDS
                // round trips of an edge (column id -> x row) hide behind arithmetic; two register sets, no rotation moves.
                int k = kbeg - q.kb4;
                const int ke = ((GML_FWABL & 1) ? kbeg : kend) - q.kb4;
                if (k < ke) {
                    struct Ops { f32x4 e[S / 4]; f32x4 t0, t1; };
                    auto fetch = [&](Ops& o, int kk, int c) {
#pragma unroll
                        for (int i = 0; i < S / 4; ++i) o.e[i] = *reinterpret_cast<const f32x4*>(ea_l + kk * S + 4 * i);
                        const int off = xoff + c * 128 + ((c & ~7) << 1);
                        o.t0 = *reinterpret_cast<const f32x4*>(lds_raw + off);
                        o.t1 = *reinterpret_cast<const f32x4*>(lds_raw + off + 16);
                    };
                    auto fma = [&](const Ops& o) {
                        const f32x2 xv[4] = {f32x2{o.t0.x, o.t0.y}, f32x2{o.t0.z, o.t0.w}, f32x2{o.t1.x, o.t1.y}, f32x2{o.t1.z, o.t1.w}};
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const float ev = o.e[s >> 2][s & 3];
                            const f32x2 e2 = f32x2{ev, ev};
#pragma unroll
                            for (int h = 0; h < 4; ++h) acc[s][h] = e2 * xv[h] + acc[s][h];
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
                if constexpr (OWNX) {
                    const int c = (int)r0 + min(row, nr - 1);
                    const int off = xoff + c * 128 + ((c & ~7) << 1);
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(lds_raw + off);
                    const f32x4 t1 = *reinterpret_cast<const f32x4*>(lds_raw + off + 16);
                    xrow[0] = t0.x; xrow[1] = t0.y; xrow[2] = t0.z; xrow[3] = t0.w;
                    xrow[4] = t1.x; xrow[5] = t1.y; xrow[6] = t1.z; xrow[7] = t1.w;
                }
            } else {                                           // group outside the LDS capacities: global gathers
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
                if constexpr (OWNX) {
                    const float* xr = p.x + (r0 + min(row, nr - 1)) * p.ldx;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        xrow[j] = (8 * kq + j < p.Fin) ? xr[8 * kq + j] : 0.f;
                        asm volatile("" : "+v"(xrow[j]));      // used here: the compiler's wait for these loads stays in this branch
                    }
                }
            }
            GML_TF3(4);

            // ---- projection: out tile = sum_s acc_s W_s (acc split on the fly = A fragments, k = f = 8*kq + j)
            f32x4 oacc[NOB];
            float oscale[NOB];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) { oacc[ob] = f32x4{0.f, 0.f, 0.f, 0.f}; oscale[ob] = 1.f; }
            if constexpr (EPL == 2) {
                // depthwise (libs/spect_conv.py:81-91): scale the aggregates per feature, add the scaled own row, project ONCE
                float hs[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) hs[j] = ds_l[S * 32 + 8 * kq + j] * xrow[j];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(ds_l + s * 32 + 8 * kq);
                    const f32x4 d1 = *reinterpret_cast<const f32x4*>(ds_l + s * 32 + 8 * kq + 4);
                    hs[0] += d0.x * acc[s][0].x; hs[1] += d0.y * acc[s][0].y; hs[2] += d0.z * acc[s][1].x; hs[3] += d0.w * acc[s][1].y;
                    hs[4] += d1.x * acc[s][2].x; hs[5] += d1.y * acc[s][2].y; hs[6] += d1.z * acc[s][3].x; hs[7] += d1.w * acc[s][3].y;
                }
                bf16x8 ah, al;
                gml_split8(hs, ah, al);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) {
                    const int o = ob * 16 + r16;
                    const int off = o * 32 + (((kq ^ gml_wkey(o)) & 3) << 3);
                    const bf16x8 wh = *reinterpret_cast<const bf16x8*>(Wof_h + off), wl = *reinterpret_cast<const bf16x8*>(Wof_l + off);
                    oacc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh, oacc[ob], 0, 0, 0);
                    oacc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl, oacc[ob], 0, 0, 0);
                    oacc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh, oacc[ob], 0, 0, 0);
                }
            } else if constexpr (EPL == 1) {
                // ConCat (libs/spect_conv.py:137-158): support s projects into its own column block s + cc_off of the output row
                const auto crs = __builtin_amdgcn_make_buffer_rsrc(p.out + r0 * p.ldo, 0, 0x7ffffe00, 0x00020000);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float av[8] = {acc[s][0].x, acc[s][0].y, acc[s][1].x, acc[s][1].y, acc[s][2].x, acc[s][2].y, acc[s][3].x, acc[s][3].y};
                    bf16x8 ah, al;
                    gml_split8(av, ah, al);
                    const int cb = (s + p.cc_off) * p.Fout;
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) {
                        const int o = ob * 16 + r16;
                        const int off = (s * 32 + o) * 32 + (((kq ^ gml_wkey(o)) & 3) << 3);
                        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(Wof_h + off), wl = *reinterpret_cast<const bf16x8*>(Wof_l + off);
                        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh, d, 0, 0, 0);
                        const float bv = (p.bias && o < p.Fout) ? p.bias[cb + o] : 0.f;
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int lr = (int)((out_rows >> (8 * reg)) & 255u);
                            const int offb = (o < p.Fout && lr < nr) ? (lr * (int)p.ldo + cb + o) * 4 : 0x7fffff00;
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d[reg] + bv), crs, offb, 0, 0);
                        }
                    }
                }
                continue;                                      // (every block is stored: nothing left for the common epilogue)
            } else {
                FT wh[2][NOB], wl[2][NOB];
                auto frag = [&](int s, int st) {
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) {
                        const int o = ob * 16 + r16;           // B[k = f][n = o]: 8 consecutive f of column o
                        const int off = (s * 32 + o) * 32 + (((kq ^ gml_wkey(o)) & 3) << 3);
                        wh[st][ob] = *reinterpret_cast<const FT*>(Wof_h + off);
                        wl[st][ob] = *reinterpret_cast<const FT*>(Wof_l + off);
                    }
                };
                frag(0, 0);
                float asc = 1.f;                               // f16 pieces: the tile's scale and the epilogue's factor per column block
                if constexpr (F16) {
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
            GML_TF3(6);
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
                if constexpr (MIX) {
                    // z[row][c] = x[row] . wmix[c]: A = the lane's row (k = f), D: lane (c = r16, kq) holds rows 4*kq + reg
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
                    const bool mine = r16 >= m0 && r16 < m0 + p.F2;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float t = F16 ? gml_tanh(fmaf(z[reg], zsc, mbias)) : gml_tanh_short(z[reg] + mbias);
                        const float u = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x128, 0xf, 0xf, true));   // row_ror:8 = lane ^ 8
                        if (mine) ov[NOB - 1][reg] = t * u;
                    }
                }
                const int lr = (int)((out_rows >> (8 * (r16 & 3))) & 255u);        // after the transpose: the row of register r16 & 3
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) {
                    gml_quad_transpose(ov[ob], lane);
                    const int c = 16 * ob + 4 * (r16 >> 2);
                    const int off = (c < ncols && lr < nr && !(GML_FWABL & 4)) ? (lr * (int)p.ldo + c) * 4 : 0x7fffff00;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov[ob]), ors, off, 0, GML_FWD3_ST_AUX);
                }
                GML_TF3(7);
            } else {
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) {
                    const int o = ob * 16 + r16;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int lr = (int)((out_rows >> (8 * reg)) & 255u);
                        float v = F16 ? fmaf(oacc[ob][reg], oscale[ob], bias_r[ob]) : oacc[ob][reg] + bias_r[ob];
                        if (relu) v = fmaxf(v, 0.f);
                        const int off = (o < p.Fout && lr < nr && !(GML_FWABL & 4)) ? (lr * (int)p.ldo + o) * 4 : 0x7fffff00;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, off, 0, 0);
                    }
                }
                GML_TF3(7);
                if constexpr (MIX) {
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
                        const float t = F16 ? gml_tanh(fmaf(z[reg], zsc, mbias)) : gml_tanh_short(z[reg] + mbias);
                        const float u = __shfl(t, lane + p.F2);    // partner column c + F2 of the same 16-lane row group
                        const int lr = (int)((out_rows >> (8 * reg)) & 255u);
                        const int off = (r16 < p.F2 && lr < nr) ? (lr * (int)p.ldo + p.mix_col + r16) * 4 : 0x7fffff00;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t * u), ors, off, 0, 0);
                    }
                }
            }
            GML_TF3(8);
        }
    }
#ifdef GML_FWD2_TIMING
    if (lane == 0 && p.prof != nullptr) {
#pragma unroll
        for (int i = 0; i < 10; ++i) atomicAdd(&p.prof[16 + i], (unsigned long long)tacc_[i]);
    }
#endif
}

template <int S, int NOB>
int gml_launch_fwd3(const GmlFwdParams& p, dim3 grid, hipStream_t st, bool mix);

#define GML_FWD3_LAUNCH_L(SV, NOBV, MX, EPV, EPLV)                                                           \
    {                                                                                                        \
        GML_ALLOW_BIG_LDS(rc_, (&gml_k_spectconv_fwd3<SV, NOBV, MX, EPV, EPLV>), 160 * 1024)                 \
        if (rc_ != hipSuccess) return (int)rc_;                                                              \
        hipLaunchKernelGGL((gml_k_spectconv_fwd3<SV, NOBV, MX, EPV, EPLV>), grid, dim3(GmlFwd3Cfg<SV>::NT),  \
                           GmlFwd3Cfg<SV>::lds_bytes(), st, p);                                              \
        return gml_launch_status();                                                                          \
    }
#define GML_FWD3_LAUNCH_F(SV, NOBV, MX, EPV, F16V)                                                           \
    {                                                                                                        \
        GML_ALLOW_BIG_LDS(rc_, (&gml_k_spectconv_fwd3<SV, NOBV, MX, EPV, 0, F16V>), 160 * 1024)              \
        if (rc_ != hipSuccess) return (int)rc_;                                                              \
        hipLaunchKernelGGL((gml_k_spectconv_fwd3<SV, NOBV, MX, EPV, 0, F16V>), grid, dim3(GmlFwd3Cfg<SV>::NT), \
                           GmlFwd3Cfg<SV>::lds_bytes(), st, p);                                              \
        return gml_launch_status();                                                                          \
    }
#define GML_FWD3_LAUNCH(SV, NOBV, MX, EPV)                                                                   \
    {                                                                                                        \
        if (p.flags & GML_F16X3) GML_FWD3_LAUNCH_F(SV, NOBV, MX, EPV, true)                                  \
        GML_FWD3_LAUNCH_F(SV, NOBV, MX, EPV, false)                                                          \
    }
#define GML_DEFINE_FWD3(SV, NOBV)                                                                            \
    template <>                                                                                              \
    int gml_launch_fwd3<SV, NOBV>(const GmlFwdParams& p, dim3 grid, hipStream_t st, bool mix) {              \
        if (p.epl != 0) {   /* ConCat / depthwise epilogues: plain SpectConv calls (no Hadamard branch, no position map) */  \
            if (mix || p.epos != nullptr) return GML_E_UNSUPPORTED;                                          \
            if (p.epl == 1) GML_FWD3_LAUNCH_L(SV, NOBV, false, false, 1)                                     \
            GML_FWD3_LAUNCH_L(SV, NOBV, false, false, 2)                                                     \
        }                                                                                                    \
        if (p.epos != nullptr) {                                                                             \
            if (mix) GML_FWD3_LAUNCH(SV, NOBV, true, true)                                                   \
            GML_FWD3_LAUNCH(SV, NOBV, false, true)                                                           \
        }                                                                                                    \
        if (mix) GML_FWD3_LAUNCH(SV, NOBV, true, false)                                                      \
        GML_FWD3_LAUNCH(SV, NOBV, false, false)                                                              \
    }
