// Fused backward, bf16x3, third layout (default for S in {2,4,6,8}, Fin <= 32, 16 < Fout <= 32, and -- NOB = 1 -- for
// counting.py's S = 12, Fout <= 16): same outputs as gml_k_spectconv_bwd / gml_k_spectconv_bwd2
//
//   dX = sum_s A_s (G W_s^T),   dval[e,s] = < X[src] W_s, G[dst] >,   dW_s = X^T (A_s G)
//
// What changed against bwd2 (profiles/r02a_*: VALU 43 %, LDS 52 % busy -- half of it bank conflicts --, matrix pipe 20 %,
// waves parked 39 % of their cycles):
//   * ONE bf16 (hi, lo) image of W in LDS, [s][o][f], serves both projections: Z^T = W^T X^T reads its A fragments with
//     ds_read_b128 (f contiguous), dX^T = W P^T reads the transposed fragments with ds_read_b64_tr_b16 (the hardware
//     transposes 4 x 16 blocks) -- 32 KB instead of 64;
//   * dX is computed transposed (D[i = f][j = row]): a lane ends up with 4 consecutive features of ITS OWN row, so the old
//     dx values and the results move as one 16-byte access per 16-wide feature block instead of 8 dword accesses to 4 rows;
//   * the row contraction of dW takes its operands from row-major bf16 images of X and P ([position][32 channels], written
//     with one ds_write_b128 per lane and image) through ds_read_b64_tr_b16: the 36 matrix-core transposes, their 72
//     conversions and 8-byte stores per tile are gone, four supports per slab (two barriers per slab, S/4 slabs);
//   * NW = 8 waves / 128-row groups (one workgroup per CU) or NW = 4 waves / 64-row groups (78 KB of LDS: two
//     independent workgroups per CU, whose matrix / LDS phases overlap each other's VALU edge phases).
// All XOR keys of the LDS images are chosen with tools/lds_sim.py (conflict-free for every access kind that touches them).
//
// NOB = 1 (Fout <= 16; 12 supports fit the registers: 2 x 12 x 4 accumulators per lane): the lane (row, kq) owns outputs
// 4 kq .. 4 kq + 3; W image [s][16 o][32 f], G window, edge loop and P image work on 16 columns.  The dX projection keeps its
// K = 32 MFMAs: k slot 8 kq + j is output 4 kq + j for j < 4, the slots j >= 4 are zero in the P fragment.
#pragma once
#include <type_traits>
#include "gml_common.h"
#include "gml_spectconv_bwd_impl.h"

// XOR key (16-byte chunks of a 64-byte row) of the W image [s][o][f]: b128 fragment reads of the natural and of the Z
// projection's permuted rows and the transposing reads of dX are all conflict-free with it
__host__ __device__ __forceinline__ int gml_wkey3(int o) { return ((o >> 3) & 1) << 1; }
// XOR key of the row-major X / P images [position][32 channels]: conflict-free ds_write_b128 (lane = (position, chunk))
// and conflict-free transposing reads of 8 consecutive positions per lane group
__host__ __device__ __forceinline__ int gml_tkey3(int pos) { return ((pos >> 2) & 1) | ((((pos >> 1) ^ (pos >> 2) ^ (pos >> 3)) & 1) << 1); }

// structural variants of the ZINC shape class (A/B builds: tools/build_variant.py <name> -DGML_B3V=<bits>):
//   1 = dval straight from the edge loop to global memory (one 8-byte store per lane and edge: no d rows in LDS, no copy-out
//       phase, the P images of slab 0 are written right behind the edge barrier -- one workgroup barrier less per group)
//   2 = the Z projection runs between the commit's LDS writes and the barrier that publishes them (it needs neither)
//   4 = the next group's loads as buffer loads from per-group descriptors: the offsets are a per-lane constant plus a scalar,
//       the hardware range check replaces the clamps (no 64-bit address arithmetic in the vector unit)
//   8 = the waves of the second half issue the next group's loads AFTER their dX projection (the first half before): the
//       address-unit-bound issue of one half overlaps the matrix-pipe-bound projection of the other (spills: not usable)
//  16 = P accumulators cleared with packed moves;  64 = s_setprio 1 for the second half of the waves
// 512 = (with 256) the NEXT group's commit (column ids, G window, row pointers -- regions dead since this group's edge barrier) sits
//       in front of the last slab barrier of this group, which publishes it: no commit barrier at the top of a group
// 1024 = (with 256) the edge loop software-pipelined over two register sets: the G row of edge k + 1 and the column id of edge
//       k + 2 are requested before the arithmetic of edge k (the loop's two dependent LDS round trips leave the critical path)
// 2048 = (with 512) the next group's loads are issued BEFORE this group's edge loop (its registers have room for them: the
//       staging registers of the rolled form are dead there): the address-unit-bound issue overlaps the VALU-bound loop
// 256 = (with 1) value rows from global memory in the edge loop (two rows in flight per lane, the group's lines touched at its
//       top): no value rows in the staging registers, in the commit or in the LDS; the freed 28 KB give the images regions of
//       their own, and with nothing aliased the barrier at the top of a group goes
// default (round 5): 1 + 4 + 16 + 256 + 512, A/B'd step by step on single boxes (profiles/r05_bwd3_variants_ab.txt)
#ifndef GML_B3V
#define GML_B3V 789
#endif

template <int S, int NFB, int NW, int NOB = 2>
struct GmlBwd3Cfg {
    static constexpr int ROWS = 16 * NW, NT = 64 * NW;
    // register-batched staging bounds (per group): 8 edges per row; 12 for counting.py's S = 12; 16 for S = 6 -- sr25.py's supports
    // have 13 entries per row (1,664 per 128 rows), and 6 supports leave the registers for it
    static constexpr int ECAP_MAX = (S > 8 ? 12 : (S == 6 ? 16 : 8)) * ROWS;
    static constexpr int XCAP_MAX = NW == 8 ? 224 : 160;
    static constexpr int LDG = 16 * NOB + 4;                 // G window rows (floats, b128 aligned)
    // NFB = 3 (33 .. 48 input features: sr25.py's 32 + 16, mutag.py's 24 + 24 hidden widths, round 5): a SECOND image of everything
    // that is laid out in 32-feature rows -- W [s][o][f 32 .. 63], the X rows -- with the same keys and access code; features >= Fin zero
    static constexpr int NIMG = NFB > 2 ? 2 : 1;
    static constexpr int W_IMG = S * 16 * NOB * 32;          // bf16 elements of one image of one half
    static constexpr int W_HALF = NIMG * W_IMG;              // bf16 elements of one (hi or lo) W image [img][s][o][32 f]
    static constexpr int W_BYTES = 2 * W_HALF * 2;
    static constexpr int SS = (S % 4 == 0) ? 4 : ((S % 3 == 0) ? 3 : S);   // supports per dW slab
    static constexpr int NSLAB = S / SS;
    static constexpr int NBLK = SS * NFB * NOB;              // 16 x 16 output blocks of a slab: (se, fb, ob)
    static constexpr int BPW = (NBLK + NW - 1) / NW;         // blocks per wave
    static constexpr int XT_IMG = 2 * ROWS * 64;             // X hi, lo   [position][32 f] of one image
    static constexpr int XT_BYTES = NIMG * XT_IMG;
    static constexpr int PT_BYTES = 2 * SS * ROWS * 64;      // P hi, lo   [se][position][32 o]
    static constexpr int GREC = 4 + ROWS / 4;                // ints per group record (ranked)
    static constexpr bool ASHARE = NW % (NOB * NFB) == 0;    // a wave's blocks share one X fragment (fb the same for all of them)
    static constexpr bool OK = (S % SS == 0) && (NFB >= 1 && NFB <= 3) && (NW == 4 || NW == 8) && (NOB == 1 || NOB == 2) &&
                               (NFB < 3 || (NOB == 2 && NW == 8 && S <= 6));
    __host__ __device__ static size_t stage_bytes(int ecap, int xcap) { return (size_t)ecap * S * 4 + (size_t)xcap * LDG * 4; }
    __host__ __device__ static size_t r_bytes(int ecap, int xcap) {          // staged values + G window, later the P slab
        const size_t a = stage_bytes(ecap, xcap);
        return a > (size_t)PT_BYTES ? a : (size_t)PT_BYTES;
    }
    // VALG (GML_B3V & 256, the ZINC shape class): the value rows are read from global memory inside the edge loop, so the LDS holds
    // column ids + G window and -- in regions of their own, aliasing nothing -- the X image and one P slab
    // (round 5: also sr25's one-launch 48-feature class, S = 6 / NFB = 3 -- its staged value rows, 24 registers, were spilling)
    static constexpr bool VALG = (GML_B3V & 256) && NOB == 2 && NW == 8 && (S == 8 || ((S == 6 || S == 4) && NFB == 3 && !(GML_B3V & 16384)));
    __host__ __device__ static size_t lds_bytes(int ecap, int xcap) {
        if (VALG) return (size_t)W_BYTES + (ROWS + 8) * 4 + (size_t)ecap * 4 + (size_t)xcap * LDG * 4 + XT_BYTES + PT_BYTES;
        return (size_t)W_BYTES + (ROWS + 8) * 4 + (size_t)ecap * 4 + r_bytes(ecap, xcap) + XT_BYTES;
    }
};

// ablation builds (tools/build_variant.py abl -DGML_ABL=<bits>): results are WRONG, the timing says what a phase costs
//   1 = every group loads the workgroup's first group (cache-hot loads), 2 = no dW phase, 4 = no edge loop,
//   8 = no dval / dx stores, 16 = no Z projection, 32 = no dX projection
#ifndef GML_ABL
#define GML_ABL 0
#endif

// timing build: every wave sums its phase durations in scalar registers (s_memtime, no memory traffic inside the loop)
// and adds them to p.prof once at the end -- the round-1 form (one global atomic per phase) shifted the waits it measured
#ifdef GML_BWD2_TIMING
#define GML_T3(i) do { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); tacc_[i] += t_ - tprev_; tprev_ = t_; } while (0)
#else
#define GML_T3(i)
#endif

// sums over the 16 lanes of a DPP row of nine values at once (butterfly: every lane ends with the totals; fixed order per lane).
// One v_add_f32_dpp per value and step, written out: the builtin form compiles to a DPP move + an add for most of them.  The nine
// chains are interleaved, so a value is read eight instructions after it was written (a DPP operand needs two wait states behind
// the VALU write, which the compiler does not insert around inline assembly: the s_nop covers the first read of a block).
#define GML_DPP9_(ctrl)                                                                                                          \
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
        "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"           \
        "v_add_f32_dpp %4, %4, %4 " ctrl " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 " ctrl " row_mask:0xf bank_mask:0xf\n\t"           \
        "v_add_f32_dpp %6, %6, %6 " ctrl " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %7, %7, %7 " ctrl " row_mask:0xf bank_mask:0xf\n\t"           \
        "v_add_f32_dpp %8, %8, %8 " ctrl " row_mask:0xf bank_mask:0xf"                                                         \
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]))
__device__ __forceinline__ void gml_row16_sum9(float (&v)[9]) {
    GML_DPP9_("quad_perm:[1,0,3,2]");
    GML_DPP9_("quad_perm:[2,3,0,1]");
    GML_DPP9_("row_half_mirror");
    GML_DPP9_("row_mirror");
}

// ablation builds of the HAD form (results WRONG, timing only): 1 = no bias sums, 2 = no own-row g loads, 4 = no Hadamard recompute
#ifndef GML_HADV
#define GML_HADV 0
#endif
// (scheduling: the stage's pieces sit around / inside the Z projection and the compiler places them -- fences around the stage and
//  explicit VALU groups between the projection's MFMA groups both measured SLOWER, profiles/r06_d_ab_notes.txt)
// LDS the HAD form adds behind the wmix rows: biases [4], dz rows [ROWS][4], per-wave bias sums [NW][36]
#define GML_BWD3_HAD_LDS(ROWS_, NW_) (16 + (ROWS_) * 16 + (NW_) * 36 * 4)

// DZ: dx starts from dz[row] . wmix (see GmlBwdParams) instead of zero / the old dx values
// HAD (with DZ; Fout = 30, two Hadamard units): the ML3Layer's output stage (libs/spect_conv.py:209-212 backward) inside this kernel --
//   dz of the group's own rows is RECOMPUTED at the top of the group from the x row the lanes hold (4 dot products over the 4 lanes
//   of a row, tanh and its derivative in the lane that ends with the sum, the partner's tanh by one permlane swap) and the row's two
//   Hadamard gradients g[row][30, 31]: neither a dz array nor a separate pass over g and x exists;
//   dw11 / dw12 = dz^T X ride in the existing row contraction: P channels 30, 31 (outputs the layer does not have) of supports 0 and
//   1 carry the four dz values, so dW_0[:, 30], dW_0[:, 31], dW_1[:, 30], dW_1[:, 31] ARE dw11[0], dw11[1], dw12[0], dw12[1];
//   the bias gradients (column sums of g over own rows, row sums of dz) are folded over the 16 rows of a wave with DPP adds and kept
//   per wave in LDS (fixed order; no registers live across the edge loop).
template <int S, int NFB, int NW, bool XV, bool DZ = false, int NOB = 2, bool HAD = false>
__global__ __launch_bounds__(64 * NW, 2) void gml_k_spectconv_bwd3(const GmlBwdParams p) {
    // (HAD without DZ: a layer whose input needs no gradient -- the model's first: no dX projection, p.dx = NULL)
    static_assert(!HAD || (XV && NOB == 2 && NFB == 2 && S >= 2), "the fused output stage is compiled for the ZINC shape class");
    using C = GmlBwd3Cfg<S, NFB, NW, NOB>;
    constexpr int NH = 2 * NOB, GC = 4 * NOB;                // f32x2 accumulators per support and lane; float4 chunks of a G row
    static_assert(!DZ || NOB == 2, "the dz hand-over is compiled for the ZINC shape class");
    constexpr int LDG = C::LDG, ROWS = C::ROWS, NT = C::NT, SS = C::SS, BPW = C::BPW;
    constexpr int VAL_ALIGN = (S % 4 == 0) ? 4 : ((S % 2 == 0) ? 2 : 1);
    constexpr bool DIRECT = (GML_B3V & 1) && NOB == 2 && (S == 8 || C::VALG);      // (the shape classes without the dval += branch)
    constexpr bool ZEARLY = (GML_B3V & 2) != 0;
    constexpr bool PINGPONG = (GML_B3V & 8) != 0;
    constexpr bool BUFLD = (GML_B3V & 4) && S == 8 && NOB == 2 && XV;
    constexpr bool PKZ = (GML_B3V & 16) != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* W_h = reinterpret_cast<__bf16*>(lds_raw);        // [s][o][f], chunks XOR gml_wkey3(o)
    __bf16* W_l = W_h + C::W_HALF;
    int* rp_l = reinterpret_cast<int*>(lds_raw + C::W_BYTES);
    int* col_l = rp_l + ROWS + 8;
    constexpr bool VALG = C::VALG;
    constexpr bool LATEC = VALG && (GML_B3V & 512) && C::NSLAB == 2;
    constexpr bool PIPE = VALG && (GML_B3V & 1024);
    // (2048 with VALG: dead -- loads return in order, the loop's first value-row wait also waits for the whole prefetch.  4096: the
    //  same placement WITHOUT value-row loads in the loop: LDS-staged value rows, only the dval stores are in flight there)
    constexpr bool EARLYI = (LATEC && (GML_B3V & 2048)) || (DIRECT && !VALG && (GML_B3V & 4096));
    static_assert(!VALG || DIRECT, "VALG needs the direct dval stores (GML_B3V bit 1)");
    unsigned char* rreg = reinterpret_cast<unsigned char*>(col_l + p.ecap);
    float* ea_l = reinterpret_cast<float*>(rreg);
    float* gs = VALG ? ea_l : ea_l + (size_t)p.ecap * S;
    // [hi, lo][position] 64-byte rows (own region);  [hi, lo][se][position] 64-byte rows (after the dval rows left; VALG: own region)
    unsigned char* xT = VALG ? rreg + (size_t)p.xcap * LDG * 4 : rreg + C::r_bytes(p.ecap, p.xcap);
    unsigned char* pT = VALG ? xT + C::XT_BYTES : rreg;
    float* wm_l = reinterpret_cast<float*>(VALG ? pT + C::PT_BYTES : xT + C::XT_BYTES);  // DZ: [4][32] rows of wmix, zero padded (the plan's lds includes these 512 bytes)
    float* hb_l = wm_l + 128;                                // HAD: b11[0], b11[1], b12[0], b12[1]
    float* dzf = hb_l + 4;                                   // HAD: [position][4] dz of the group's rows
    float* bsum = dzf + ROWS * 4;                            // HAD: [wave][36]: column sums of g (0 .. 31), row sums of dz (32 .. 35)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);

    if constexpr ((GML_B3V & 64) != 0) {
        if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
    }
#ifdef GML_B3DELAY
    // experiment (NW = 4, two workgroups per CU): two workgroups started together run IN PHASE (same code, same durations: both in
    // their VALU-bound edge loop, then both in their matrix-pipe phases); delaying one of each pair by ~half a group puts one's
    // edge loop beside the other's projections.  GML_B3DELAY = number of s_sleep(127) (8 k cycles each), GML_B3DELAY_ODD: which half
    if constexpr (NW == 4) {
#ifdef GML_B3DELAY_ODD
        const bool late = (blockIdx.x & 1) != 0;
#else
        const bool late = blockIdx.x >= gridDim.x / 2;
#endif
        if (late)
            for (int i = 0; i < GML_B3DELAY; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    if constexpr (DZ || HAD) {
        if (tid < 128) wm_l[tid] = ((tid >> 5) < p.nmix && (tid & 31) < p.Fin) ? ((tid >> 5) < p.nmix1 ? p.wmix[(tid >> 5) * p.Fin + (tid & 31)] : p.wmix2[((tid >> 5) - p.nmix1) * p.Fin + (tid & 31)]) : 0.f;
    }
    if constexpr (HAD) {
        if (tid < 4) hb_l[tid] = tid < 2 ? (p.hb11 ? p.hb11[tid] : 0.f) : (p.hb12 ? p.hb12[tid - 2] : 0.f);
        if (tid < NW * 36) bsum[tid] = 0.f;
    }
    // W -> bf16 (hi, lo) image, zero padded to 32 x 32
    constexpr int WO = 16 * NOB;                             // output rows of one support's image
    constexpr int NIMG = C::NIMG;
    for (int e = tid; e < NIMG * S * WO * 32; e += NT) {
        const int fl = e & 31, o = (e >> 5) % WO, s = ((e >> 5) / WO) % S, img = (e >> 5) / (WO * S);
        const int f = 32 * img + fl;
        const float v = (f < p.Fin && o < p.Fout) ? p.w[((int64_t)s * p.Fin + f) * p.Fout + o] : 0.f;
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        const int i = img * C::W_IMG + (s * WO + o) * 32 + ((((fl >> 3) ^ gml_wkey3(o)) & 3) << 3) + (fl & 7);
        W_h[i] = h; W_l[i] = l;
    }

    f32x4 dwacc[C::NSLAB][BPW];
#pragma unroll
    for (int i = 0; i < C::NSLAB; ++i)
#pragma unroll
        for (int j = 0; j < BPW; ++j) dwacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef GML_BWD2_TIMING
    unsigned tacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tprev_ = (unsigned)__builtin_readcyclecounter();
#endif
    // ---- staging registers: a group's global loads are issued one phase early (after the previous group's edge phase,
    //      before its stores) and committed to LDS at the top of the group; all unconditional with clamped indices so that
    //      the compiler can count them (see gml_spectconv_bwd2_impl.h)
    // staged value rows move as float4 (S % 4 == 0) or float2 (S = 6, 2: 8-byte aligned rows) -- round 4: before, only float4 rows were
    // prefetched and a 6-support group staged with plain global -> LDS copies at its top (sr25: the whole load latency exposed per group)
    constexpr int VW = (S % 4 == 0) ? 4 : ((S % 2 == 0) ? 2 : 1);
    typedef typename std::conditional<VW == 4, f32x4, f32x2>::type EV;
    constexpr int NC = C::ECAP_MAX / NT, NE4 = (VW > 1) ? C::ECAP_MAX * (S / VW) / NT : 1;
    constexpr int NG4 = (C::XCAP_MAX * GC + NT - 1) / NT;
    const int etot = p.rowptr[p.nrows];
    const int* colb = etot > 0 ? p.col : p.ginfo;
    const EV* valb = etot > 0 ? reinterpret_cast<const EV*>(p.val) : reinterpret_cast<const EV*>(p.ginfo);
    const int emax = max(etot, 1) - 1;
    const int64_t emax4 = (VW > 1) ? max((int64_t)etot * (S / VW), (int64_t)1) - 1 : 0;
    const int o4max = p.gvec ? ((p.Fout + 3) / 4 * 4 - 4) : 0;
    int cv[NC], rpv = 0, row_n = 0;
    EV ev4[NE4];
    f32x4 gv4[NG4];
    float xb[8];
    float xb2[NIMG == 2 ? 8 : 1];                            // NFB = 3: features 32 + 8 kq .. + 7 of the own row (kq >= 2: beyond 48, zero)
    float gown[HAD ? 8 : 1];                                 // HAD: columns 8 kq .. + 7 of the own row of g
    auto vec_group = [&](const int4 gi) {
        return (VW > 1) && p.gvec && gi.y <= C::ECAP_MAX && gi.w <= C::XCAP_MAX;
    };
    int4 gi_nv = int4{0, 0, 0, 0};
    int4 gi_c = int4{0, 0, 0, 0};
    auto latch = [&]() {
        gi_c = int4{__builtin_amdgcn_readfirstlane(gi_nv.x), __builtin_amdgcn_readfirstlane(gi_nv.y),
                    __builtin_amdgcn_readfirstlane(gi_nv.z), __builtin_amdgcn_readfirstlane(gi_nv.w)};
    };
    const int voff_g = (tid / GC) * (int)p.ldg * 4 + (tid % GC) * 16;   // BUFLD: the lane's offset inside the G window (chunk tid of trip 0)
    auto issue = [&](int g) {
        const int64_t r0 = (int64_t)g * ROWS;
        const int4 gi = gi_c;
        const int kb = gi.x, ne = gi.y, lo = gi.z, nwin = gi.w;
        if constexpr (BUFLD) {
            // descriptors based at the group's first row pointer / edge / window row / x row: every offset is small, whatever
            // the size of the arrays; elements past the arrays read zeros, elements past the group are never committed
            auto rsrc = [](const void* base, int64_t nbytes) {
                return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(nbytes < 0 ? 0 : (nbytes > 0x7fffff00 ? 0x7fffff00 : nbytes)), 0x00020000);
            };
            const int64_t rem = (int64_t)etot - kb;
            const auto rs_rp = rsrc(p.rowptr + r0, (p.nrows + 1 - r0) * 4);
            const auto rs_col = rsrc(p.col + kb, rem * 4);
            const auto rs_val = rsrc(p.val + (int64_t)kb * S, rem * (S * 4));
            const auto rs_g = rsrc(p.g + (int64_t)lo * p.ldg, (p.nrows - lo) * p.ldg * 4);
            const auto rs_x = rsrc(p.x + r0 * p.ldx, (p.nrows - r0) * p.ldx * 4);
            const int ldxb = (int)p.ldx * 4, ldgb = (int)p.ldg * 4;
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_x, row_n * ldxb + 32 * kq + 16 * q4, 0, 0);
                xb[4 * q4] = __uint_as_float(t.x); xb[4 * q4 + 1] = __uint_as_float(t.y);
                xb[4 * q4 + 2] = __uint_as_float(t.z); xb[4 * q4 + 3] = __uint_as_float(t.w);
            }
            if constexpr (HAD && !(GML_HADV & 2)) {
                const auto rs_go = rsrc(p.g + r0 * p.ldg, (p.nrows - r0) * p.ldg * 4);
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_go, row_n * ldgb + 32 * kq + 16 * q4, 0, 0);
                    gown[4 * q4] = __uint_as_float(t.x); gown[4 * q4 + 1] = __uint_as_float(t.y);
                    gown[4 * q4 + 2] = __uint_as_float(t.z); gown[4 * q4 + 3] = __uint_as_float(t.w);
                }
            }
            rpv = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_rp, tid * 4, 0, 0);
#pragma unroll
            for (int t = 0; t < NC; ++t) cv[t] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_col, tid * 4, NT * 4 * t, 0);
#pragma unroll
            for (int t = 0; t < (VALG ? 0 : NE4); ++t) ev4[t] = __builtin_bit_cast(EV, __builtin_amdgcn_raw_buffer_load_b128(rs_val, tid * 16, NT * 16 * t, 0));
#pragma unroll
            for (int t = 0; t < NG4; ++t) gv4[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, voff_g, t * (NT / GC) * ldgb, 0));
            return;
        }
        if constexpr (XV) {
            const float* xr = p.x + min(r0 + row_n, p.nrows - 1) * p.ldx;
            const int f4max = (p.Fin + 3) / 4 * 4 - 4;
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(xr + min(8 * kq + 4 * q4, f4max));
                xb[4 * q4] = t.x; xb[4 * q4 + 1] = t.y; xb[4 * q4 + 2] = t.z; xb[4 * q4 + 3] = t.w;
                if constexpr (NIMG == 2) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(xr + min(32 + 8 * kq + 4 * q4, f4max));
                    xb2[4 * q4] = u.x; xb2[4 * q4 + 1] = u.y; xb2[4 * q4 + 2] = u.z; xb2[4 * q4 + 3] = u.w;
                }
            }
        }
        if constexpr (HAD && !(GML_HADV & 2)) {
            const float* gr = p.g + min(r0 + row_n, p.nrows - 1) * p.ldg + 8 * kq;
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(gr + 4 * q4);
                gown[4 * q4] = t.x; gown[4 * q4 + 1] = t.y; gown[4 * q4 + 2] = t.z; gown[4 * q4 + 3] = t.w;
            }
        }
        rpv = p.rowptr[min(r0 + tid, p.nrows)];
        const int ne1 = max(ne, 1) - 1, ne41 = max(ne * (S / (VW > 1 ? VW : 1)), 1) - 1, nw1 = max(nwin, 1) - 1;
#pragma unroll
        for (int t = 0; t < NC; ++t) cv[t] = colb[min(kb + min(tid + NT * t, ne1), emax)];
#pragma unroll
        for (int t = 0; t < (VALG ? 0 : NE4); ++t) ev4[t] = valb[min((int64_t)kb * (S / (VW > 1 ? VW : 1)) + min(tid + NT * t, ne41), emax4)];
#pragma unroll
        for (int t = 0; t < NG4; ++t) {
            const int i = tid + NT * t;
            const int64_t rr = min((int64_t)lo + min(i / GC, nw1), p.nrows - 1);
            gv4[t] = *reinterpret_cast<const f32x4*>(p.g + rr * p.ldg + min((i % GC) * 4, o4max));
        }
    };
    auto load_rows = [&](int g) {                            // ranked record: {kb, ne, lo, nwin}, then the row of every position
        const int32_t* rec = p.ginfo + (int64_t)g * C::GREC;
        gi_nv = *reinterpret_cast<const int4*>(rec);
        row_n = reinterpret_cast<const unsigned char*>(rec + 4)[wave * 16 + r16];
    };
    auto commit = [&](const int4 gi, const int nr) {
        const int kb = gi.x, ne = gi.y, lo = gi.z, nwin = gi.w;
        if (tid <= nr) rp_l[tid] = rpv;
        if (vec_group(gi)) {
#pragma unroll
            for (int t = 0; t < NC; ++t) { const int i = tid + NT * t; if (i < ne) col_l[i] = cv[t] - lo; }
#pragma unroll
            for (int t = 0; t < (VALG ? 0 : NE4); ++t) {
                const int i = tid + NT * t;
                if (i < ne * (S / (VW > 1 ? VW : 1))) reinterpret_cast<EV*>(ea_l)[i] = ev4[t];
            }
#pragma unroll
            for (int t = 0; t < NG4; ++t) {
                const int i = tid + NT * t;
                if (i < nwin * GC)
                    *reinterpret_cast<f32x4*>(gs + (i / GC) * LDG + (i % GC) * 4) = ((i % GC) * 4 < p.Fout) ? gv4[t] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
            for (int i = tid; i < ne; i += NT) col_l[i] = p.col[kb + i] - lo;
            if constexpr (!VALG)
                for (int i = tid; i < ne * S; i += NT) ea_l[i] = p.val[(int64_t)kb * S + i];
            for (int i = tid; i < nwin * 16 * NOB; i += NT) {
                const int rr = i / (16 * NOB), o = i % (16 * NOB);
                gs[rr * LDG + o] = (o < p.Fout) ? p.g[(int64_t)(lo + rr) * p.ldg + o] : 0.f;
            }
        }
    };
    if (g0 < g1) { load_rows(g0); latch(); issue(g0); }
    if constexpr (LATEC) {                                   // the first group's commit (every later one: inside the previous group)
        if (g0 < g1) commit(gi_c, (int)min((int64_t)ROWS, p.nrows - (int64_t)g0 * ROWS));
        __syncthreads();
    }
    if constexpr (HAD && !LATEC) __syncthreads();           // wm_l / hb_l / bsum are read at the top of the first group
    for (int g = g0; g < g1; ++g) {
        const int64_t r0 = (int64_t)g * ROWS;
        const int nr = (int)min((int64_t)ROWS, p.nrows - r0);
        const int4 gi = gi_c;
        const int kb = gi.x, ne = gi.y, lo = gi.z, nwin = gi.w;
        const int row = row_n;
        load_rows((GML_ABL & 1) ? g0 : min(g + 1, g1 - 1));
        // value rows of THIS group (VALG): one descriptor for the edge loop's loads; every 128-byte line of the group is touched now --
        // a commit and a Z projection ahead of its first use -- so that the loop's loads find it in the L2 / L1
        const auto vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(VALG ? p.val + (int64_t)kb * S : p.x), 0, VALG ? ne * (S * 4) : 0, 0x00020000);
        uint32_t vtouch = 0;
        if constexpr (VALG) vtouch = __builtin_amdgcn_raw_buffer_load_b32(vrs, tid * 128, 0, 0);
        // (VALG: the images live in regions nothing else uses, and whoever writes them has passed this group's commit barrier, which
        //  every wave reaches with the previous group's contraction behind it: no barrier here)
        if constexpr (!VALG) __syncthreads();                // previous group is done with every LDS region
        GML_T3(0);

        // ---- stage: commit the registers loaded one phase ago
        const bool rvalid = row < nr;
        if constexpr (!XV) {
            const float* xr = p.x + (r0 + row) * p.ldx + 8 * kq;
#pragma unroll
            for (int t = 0; t < 8; ++t) xb[t] = (rvalid && 8 * kq + t < p.Fin) ? xr[t] : 0.f;
            if constexpr (NIMG == 2) {
#pragma unroll
                for (int t = 0; t < 8; ++t) xb2[t] = (rvalid && 32 + 8 * kq + t < p.Fin) ? xr[32 + t] : 0.f;
            }
        }
        f32x4 dzv = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- HAD: output stage of the group's own rows, in three pieces inside the Z projection's code (its matrix-pipe chain leaves
        //      most VALU issue slots free; the compiler interleaves): (a) LDS reads + the four partial dot products, (b) pure VALU:
        //      fold over the row's lanes, tanh, dz, the DPP sums of the bias gradients, (c) the per-wave records + dz back as one float4
        float had_a[4], had_g6 = 0.f, had_g7 = 0.f, had_bs[9];   // (had_bs: column sums of g, 0 .. 7; the row sum of dz, 8)
        const int pos_h = wave * 16 + r16;
        auto had_pre = [&]() {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (!(rvalid && 8 * kq + j < p.Fin)) xb[j] = 0.f;
                if ((!BUFLD && !rvalid) || (GML_HADV & 2)) gown[j] = 0.f;       // (buffer loads: rows beyond the array read zeros)
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wm_l + q * 32 + 8 * kq);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wm_l + q * 32 + 8 * kq + 4);
                f32x2 a2 = f32x2{xb[0], xb[1]} * f32x2{w0.x, w0.y};
                a2 = f32x2{xb[2], xb[3]} * f32x2{w0.z, w0.w} + a2;
                a2 = f32x2{xb[4], xb[5]} * f32x2{w1.x, w1.y} + a2;
                a2 = f32x2{xb[6], xb[7]} * f32x2{w1.z, w1.w} + a2;
                had_a[q] = a2.x + a2.y;
            }
            had_g6 = __shfl(gown[6], r16 + 48); had_g7 = __shfl(gown[7], r16 + 48);   // g[row][30], g[row][31] sit in lane kq = 3
        };
        auto had_valu = [&]() {
            // the four lanes of a row fold their partial sums: lane kq ends with pre-activation kq (fc11 units 0, 1, fc12 units 0, 1)
            const auto a01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(had_a[0]), __float_as_uint(had_a[1]), false, false);
            const auto a23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(had_a[2]), __float_as_uint(had_a[3]), false, false);
            const float c01 = __uint_as_float(a01[0]) + __uint_as_float(a01[1]);
            const float c23 = __uint_as_float(a23[0]) + __uint_as_float(a23[1]);
            const auto bq = __builtin_amdgcn_permlane32_swap(__float_as_uint(c01), __float_as_uint(c23), false, false);
            const float pre = __uint_as_float(bq[0]) + __uint_as_float(bq[1]) + hb_l[kq];
            float th, dh;
            gml_tanh_d(pre, th, dh);
            const auto tsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(th), __float_as_uint(th), false, false);
            const float tpart = __uint_as_float(kq < 2 ? tsw[1] : tsw[0]);       // tanh of the partner unit (lane kq ^ 2)
            const float dzo = (GML_HADV & 4) ? 0.f : ((kq & 1) ? had_g7 : had_g6) * tpart * dh;   // dz[row][kq]
            dzf[pos_h * 4 + kq] = dzo;
            if constexpr (!(GML_HADV & 1)) {                 // bias gradients: sums over the wave's 16 rows (every lane ends with them)
#pragma unroll
                for (int j = 0; j < 8; ++j) had_bs[j] = gown[j];
                had_bs[8] = dzo;
                gml_row16_sum9(had_bs);
            }
        };
        auto had_post = [&]() {
            if constexpr (!(GML_HADV & 1)) {
                if (r16 == 0) {                              // one lane per kq adds them to the wave's record
                    float* bw = bsum + wave * 36;
                    f32x4 o0 = *reinterpret_cast<const f32x4*>(bw + 8 * kq), o1 = *reinterpret_cast<const f32x4*>(bw + 8 * kq + 4);
                    o0 += f32x4{had_bs[0], had_bs[1], had_bs[2], had_bs[3]};
                    o1 += f32x4{had_bs[4], had_bs[5], had_bs[6], had_bs[7]};
                    *reinterpret_cast<f32x4*>(bw + 8 * kq) = o0;
                    *reinterpret_cast<f32x4*>(bw + 8 * kq + 4) = o1;
                    bw[32 + kq] += had_bs[8];
                }
            }
            dzv = *reinterpret_cast<const f32x4*>(dzf + pos_h * 4);              // (the wave's own writes: no barrier)
        };
        if constexpr (!LATEC) commit(gi, nr);
        bf16x8 xh, xl;                                       // own X row, features 8*kq .. 8*kq+7: B fragment of Z^T, row of the X image
        bf16x8 xh2, xl2;                                     // NFB = 3: features 32 + 8*kq .. + 7 (the second K block / image)
        unsigned xpos = 0;                                   // DZ, relu_cols > 0: bit j = (x[row][8 kq + j] > 0), the relu mask of the layer below
        f32x2 Z[S][NH], P[S][NH];
        auto zproj = [&]() {
        if constexpr (HAD) had_pre();
        if constexpr (XV) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!(rvalid && 8 * kq + j < p.Fin)) xb[j] = 0.f;
        }
        gml_split8(xb, xh, xl);
        if constexpr (NIMG == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!(rvalid && 32 + 8 * kq + j < p.Fin)) xb2[j] = 0.f;
            gml_split8(xb2, xh2, xl2);
        }
        if constexpr (DZ) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xpos |= (xb[j] > 0.f ? 1u : 0u) << j;
        }

        // ---- Z^T = W^T X^T: MFMA row i of block ob carries o = 8*(i>>2) + 4*ob + (i&3), so lane kq receives its 8
        //      consecutive outputs o = 8*kq + 4*ob + reg.  Fragments of support s + 1 are requested before the MFMAs of s.
        {
            const int oa0 = NOB == 2 ? 8 * (r16 >> 2) + (r16 & 3) : r16;   // (NOB = 1: MFMA row i = output i, lane kq receives 4 kq + reg)
            bf16x8 wh[2][NOB], wl[2][NOB];
            bf16x8 wh2[2][NIMG == 2 ? NOB : 1], wl2[2][NIMG == 2 ? NOB : 1];
            auto frag = [&](int s, int st) {
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) {
                    const int oa = oa0 + 4 * ob;
                    const int off = (s * WO + oa) * 32 + (((kq ^ gml_wkey3(oa)) & 3) << 3);
                    wh[st][ob] = *reinterpret_cast<const bf16x8*>(W_h + off);
                    wl[st][ob] = *reinterpret_cast<const bf16x8*>(W_l + off);
                    if constexpr (NIMG == 2) {
                        wh2[st][ob] = *reinterpret_cast<const bf16x8*>(W_h + C::W_IMG + off);
                        wl2[st][ob] = *reinterpret_cast<const bf16x8*>(W_l + C::W_IMG + off);
                    }
                }
            };
            if (!(GML_ABL & 16)) frag(0, 0);
            if constexpr (HAD) had_valu();                   // (no LDS reads inside: the pipeline below counts them)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int st = s & 1;
                if (GML_ABL & 16) {
#pragma unroll
                    for (int h = 0; h < NH; ++h) { Z[s][h] = f32x2{xb[h], xb[h + 4]}; P[s][h] = f32x2{0.f, 0.f}; }
                    continue;
                }
                if (s + 1 < S) frag(s + 1, st ^ 1);
                f32x4 dd[NOB];
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) dd[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) dd[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[st][ob], xh, dd[ob], 0, 0, 0);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) dd[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[st][ob], xl, dd[ob], 0, 0, 0);
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) dd[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[st][ob], xh, dd[ob], 0, 0, 0);
                if constexpr (NIMG == 2) {
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) dd[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl2[st][ob], xh2, dd[ob], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) dd[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh2[st][ob], xl2, dd[ob], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) dd[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh2[st][ob], xh2, dd[ob], 0, 0, 0);
                }
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) { Z[s][2 * ob] = f32x2{dd[ob][0], dd[ob][1]}; Z[s][2 * ob + 1] = f32x2{dd[ob][2], dd[ob][3]}; }
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    if constexpr (PKZ) asm volatile("v_pk_mov_b32 %0, 0, 0" : "=v"(P[s][h]));
                    else P[s][h] = f32x2{0.f, 0.f};
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NOB * NIMG, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NOB * NIMG, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NOB * NIMG, 0);
            }
        }
        if constexpr (HAD) had_post();
        };
        if constexpr (ZEARLY) zproj();                       // (needs the W image and the lane's own x row only: neither is part of the commit)
        if constexpr (!LATEC) __syncthreads();
        GML_T3(1);
        // old dx values (accumulate mode): the lane's own row, features 16 fb + 4 kq .. + 3 (D rows of dX^T);
        // DZ: the row's 4 Hadamard-branch gradients instead (one 16-byte load; dx then starts from dz . wmix)
        f32x4 dxa[NFB];
        const bool dxv = p.dx && p.dxvec;                    // dx rows float4-addressable (Fin % 4 == 0, aligned rows)
        if constexpr (DZ) {
            if constexpr (!HAD) dzv = *reinterpret_cast<const f32x4*>(p.dz + min(r0 + row, p.nrows - 1) * 4);
        } else if constexpr (!HAD) {
            const float* dxb = p.dx ? p.dx : p.x;            // (no dx wanted: any readable rows, the values are dropped)
            const int64_t ldb = p.dx ? p.lddx : p.ldx;
            const float* dr = dxb + min(r0 + row, p.nrows - 1) * ldb;
            if (dxv) {
                const int f4max = (p.Fin + 3) / 4 * 4 - 4;
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = *reinterpret_cast<const f32x4*>(dr + min(16 * fb + 4 * kq, f4max));
            } else {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) dxa[fb][reg] = dr[min(16 * fb + 4 * kq + reg, p.Fin - 1)];
            }
        }

        const int kbeg = rvalid ? rp_l[row] - kb : 0;
        const int kend = rvalid ? rp_l[row + 1] - kb : 0;

        if constexpr (!ZEARLY) zproj();
        if constexpr (EARLYI) {
            latch();
            issue((GML_ABL & 1) ? g0 : min(g + 1, g1 - 1));
        }

        GML_T3(2);
        // ---- edge phase (fp32 VALU, packed): P += val * G[dst],  d[s] = <Z[s], G[dst]>
        // (tried and measured slower, r02: requesting edge k + 1's value row / G row and edge k + 2's column before the
        //  arithmetic of edge k -- the rotation costs ~20 moves and two clamps per trip and pushes 16 registers into spills)
        // DIRECT: dval[kb + k][2 kq, 2 kq + 1] leaves the loop as one 8-byte store (a buffer based at the group's first value row; no
        // dval wanted: zero records, the hardware drops the stores)
        const auto dvrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DIRECT && p.dval ? p.dval + (int64_t)kb * S : p.x), 0,
                                                            (DIRECT && p.dval && !(GML_ABL & 8)) ? ne * (S * 4) : 0, 0x00020000);
        auto ldg_row = [&](int dstl, f32x2 (&gv)[NH]) {      // the lane's 4 NOB columns of the destination's G row
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(gs + dstl * LDG + 4 * NOB * kq + 4 * ob);
                gv[2 * ob] = f32x2{t0.x, t0.y}; gv[2 * ob + 1] = f32x2{t0.z, t0.w};
            }
        };
        auto edge_g = [&](int k, const float (&ev)[S], const f32x2 (&gv)[NH]) {
            float d[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f32x2 a2 = f32x2{0.f, 0.f};
                const f32x2 e2 = f32x2{ev[s], ev[s]};
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    P[s][h] = e2 * gv[h] + P[s][h];
                    a2 = Z[s][h] * gv[h] + a2;
                }
                d[s] = a2.x + a2.y;
            }
            if constexpr (DIRECT && S == 4) {                // four supports: lane kq ends with support kq (one dword store)
                const auto a01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[0]), __float_as_uint(d[1]), false, false);
                const auto a23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[2]), __float_as_uint(d[3]), false, false);
                const float c01 = __uint_as_float(a01[0]) + __uint_as_float(a01[1]);
                const float c23 = __uint_as_float(a23[0]) + __uint_as_float(a23[1]);
                const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(c01), __float_as_uint(c23), false, false);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__uint_as_float(b[0]) + __uint_as_float(b[1])), dvrs, (k * S + kq) * 4, 0, 0);
            } else if constexpr (DIRECT) {                   // fold slot j of chunk c = support 2 j + c: lane kq ends with supports 2 kq, 2 kq + 1
                float tot2[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float d6 = (S > 6) ? d[(S > 6) ? 6 + c : 0] : 0.f;       // (S = 6: slots 6, 7 are empty -- lane kq = 3 stores nothing)
                    const auto a01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[c]), __float_as_uint(d[2 + c]), false, false);
                    const auto a23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[4 + c]), __float_as_uint(d6), false, false);
                    const float c01 = __uint_as_float(a01[0]) + __uint_as_float(a01[1]);
                    const float c23 = __uint_as_float(a23[0]) + __uint_as_float(a23[1]);
                    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(c01), __float_as_uint(c23), false, false);
                    tot2[c] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
                }
                typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
                // (S = 6: the offset of lane kq = 3 lies beyond every record: the hardware drops its store)
                const int doff = (S == 8 || kq < S / 2) ? (k * S + 2 * kq) * 4 : 0x7ffffff0;
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_{__float_as_uint(tot2[0]), __float_as_uint(tot2[1])}, dvrs, doff, 0, 0);
            } else {
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
        };
        auto edge = [&](int k, const float (&ev)[S]) {
            f32x2 gv[NH];
            ldg_row(col_l[k], gv);
            edge_g(k, ev, gv);
        };
        if constexpr (PIPE) {
            asm volatile("" :: "v"(vtouch));
            auto ldval = [&](int k, float (&ev)[S]) {
                const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(vrs, k * (S * 4), 0, 0);
                ev[0] = __uint_as_float(a.x); ev[1] = __uint_as_float(a.y); ev[2] = __uint_as_float(a.z); ev[3] = __uint_as_float(a.w);
                if constexpr (S == 8) {
                    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(vrs, k * (S * 4) + 16, 0, 0);
                    ev[4] = __uint_as_float(b.x); ev[5] = __uint_as_float(b.y); ev[6] = __uint_as_float(b.z); ev[7] = __uint_as_float(b.w);
                } else if constexpr (S == 6) {               // 24-byte rows
                    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
                    const u32x2v b = __builtin_amdgcn_raw_buffer_load_b64(vrs, k * (S * 4) + 16, 0, 0);
                    ev[4] = __uint_as_float(b.x); ev[5] = __uint_as_float(b.y);
                }
            };
            int k = kbeg;
            const int klast = kend - 1;
            if (k < ((GML_ABL & 4) ? kbeg : kend)) {
                float eA[S], eB[S];
                f32x2 gA[NH], gB[NH];
                ldval(k, eA);
                ldg_row(col_l[k], gA);
                int cn = col_l[min(k + 1, klast)];
                for (;;) {
                    const int c2 = col_l[min(k + 2, klast)];
                    ldval(min(k + 1, klast), eB);
                    ldg_row(cn, gB);
                    edge_g(k, eA, gA);
                    if (++k >= kend) break;
                    cn = col_l[min(k + 2, klast)];
                    ldval(min(k + 1, klast), eA);
                    ldg_row(c2, gA);
                    edge_g(k, eB, gB);
                    if (++k >= kend) break;
                }
            }
        } else if constexpr (VALG) {
            // two value rows in flight per lane (two register sets, no rotation): the row of edge k + 1 is requested before the
            // arithmetic of edge k
            asm volatile("" :: "v"(vtouch));
            auto ldval = [&](int k, float (&ev)[S]) {
                const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(vrs, k * (S * 4), 0, 0);
                ev[0] = __uint_as_float(a.x); ev[1] = __uint_as_float(a.y); ev[2] = __uint_as_float(a.z); ev[3] = __uint_as_float(a.w);
                if constexpr (S == 8) {
                    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(vrs, k * (S * 4) + 16, 0, 0);
                    ev[4] = __uint_as_float(b.x); ev[5] = __uint_as_float(b.y); ev[6] = __uint_as_float(b.z); ev[7] = __uint_as_float(b.w);
                } else if constexpr (S == 6) {               // 24-byte rows
                    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
                    const u32x2v b = __builtin_amdgcn_raw_buffer_load_b64(vrs, k * (S * 4) + 16, 0, 0);
                    ev[4] = __uint_as_float(b.x); ev[5] = __uint_as_float(b.y);
                }
            };
            int k = kbeg;
            const int klast = kend - 1;
            if (k < ((GML_ABL & 4) ? kbeg : kend)) {
                float eA[S], eB[S];
                ldval(k, eA);
                if constexpr ((GML_B3V & 8192) != 0) {
                    // the column id of edge k + 1 travels one trip ahead (one register, one clamp): the loop's chain column id -> G row ->
                    // arithmetic loses its first LDS round trip (the full two-set pipeline of G rows costs more than it hides: bit 1024)
                    auto edge_c = [&](int kk, int c, const float (&ev)[S]) {
                        f32x2 gv[NH];
                        ldg_row(c, gv);
                        edge_g(kk, ev, gv);
                    };
                    int c = col_l[k];
                    for (;;) {
                        const int kn = min(k + 1, klast);
                        const int cn = col_l[kn];
                        ldval(kn, eB);
                        edge_c(k, c, eA);
                        if (++k >= kend) break;
                        const int kn2 = min(k + 1, klast);
                        c = col_l[kn2];
                        ldval(kn2, eA);
                        edge_c(k, cn, eB);
                        if (++k >= kend) break;
                    }
                } else if constexpr ((GML_B3V & 32768) != 0) {
                    // pairs: the G rows of edges k and k + 1 are both requested at the top of a two-edge trip (two fixed register sets,
                    // no rotation): the second gather's LDS round trip runs behind the first edge's arithmetic
                    for (;;) {
                        f32x2 gA[NH], gB[NH];
                        const int kn = min(k + 1, klast);
                        ldg_row(col_l[k], gA);
                        ldg_row(col_l[kn], gB);
                        ldval(kn, eB);
                        edge_g(k, eA, gA);
                        if (++k >= kend) break;
                        ldval(min(k + 1, klast), eA);
                        edge_g(k, eB, gB);
                        if (++k >= kend) break;
                    }
                } else {
                for (;;) {
                    ldval(min(k + 1, klast), eB);
                    edge(k, eA);
                    if (++k >= kend) break;
                    ldval(min(k + 1, klast), eA);
                    edge(k, eB);
                    if (++k >= kend) break;
                }
                }
            }
        } else {
            for (int k = kbeg; k < ((GML_ABL & 4) ? kbeg : kend); ++k) {
                float ev[S];
                gml_load_row<S, VAL_ALIGN>(ea_l + k * S, ev);
                edge(k, ev);
            }
        }
        GML_T3(3);
        __syncthreads();                                     // dval rows complete; G window no longer needed
        GML_T3(4);
        // Lane-only address terms of the phases below are recomputed per group from an opaque copy of the thread id: left
        // visible, they are hoisted out of the group loop, spilled, and every reload waits (vmcnt(0)) for the whole
        // prefetch that was just issued -- measured as 16 % of the kernel in front of the dval stores.
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const int r16o = tid_o & 15, kqo = (tid_o >> 4) & 3, waveo = __builtin_amdgcn_readfirstlane(tid_o >> 6);
        // P -> bf16 (hi, lo) once: B fragments of dX^T (k = o = 8*kq + j) and the rows of the P image.  Before the next
        // group's loads are issued: P and its split are both live here, the prefetch registers are not yet.
        typedef typename std::conditional<NOB == 2, bf16x8, bf16x4>::type PFrag;   // (NOB = 1: the 4 live k slots; the rest is zero)
        if constexpr (HAD) {                                 // channels 30, 31 of supports 0, 1 carry dz (their W rows are zero: dX is not touched)
            if (kqo == 3) {
                P[0][3] = f32x2{dzv[0], dzv[1]};
                P[1][3] = f32x2{dzv[2], dzv[3]};
            }
        }
        PFrag PH[S], PL[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float pv[8];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f32x2 t = h < NH ? P[s][h < NH ? h : 0] : f32x2{0.f, 0.f};
                pv[2 * h] = t.x; pv[2 * h + 1] = t.y;
            }
            bf16x8 th, tl;
            gml_split8(pv, th, tl);                          // (NOB = 1: k slots 8 kq + 4 .. + 7 are zero, like their W rows)
            if constexpr (NOB == 2) { PH[s] = th; PL[s] = tl; }
            else { PH[s] = bf16x4{th[0], th[1], th[2], th[3]}; PL[s] = bf16x4{tl[0], tl[1], tl[2], tl[3]}; }
        }
        GML_T3(10);
        // row-major bf16 images [position = wave*16 + r16][32 channels] of X and P for the dW contraction (see below)
        const int pos = waveo * 16 + r16o;
        const int woff = pos * 64 + (((kqo ^ gml_tkey3(pos)) & 3) << 4);
        auto write_x = [&]() {
            *reinterpret_cast<bf16x8*>(xT + woff) = xh;
            *reinterpret_cast<bf16x8*>(xT + ROWS * 64 + woff) = xl;
            if constexpr (NIMG == 2) {
                *reinterpret_cast<bf16x8*>(xT + C::XT_IMG + woff) = xh2;
                *reinterpret_cast<bf16x8*>(xT + C::XT_IMG + ROWS * 64 + woff) = xl2;
            }
        };
        auto write_slab = [&](int sl) {
#pragma unroll
            for (int se = 0; se < SS; ++se) {
                const int s = sl * SS + se;
                if constexpr (NOB == 2) {
                    *reinterpret_cast<bf16x8*>(pT + se * ROWS * 64 + woff) = PH[s];
                    *reinterpret_cast<bf16x8*>(pT + (SS + se) * ROWS * 64 + woff) = PL[s];
                } else {                                     // compact: channel = output 4 kq + j = 8-byte piece kq of the row
                    const int woff1 = pos * 64 + ((((kqo >> 1) ^ gml_tkey3(pos)) & 3) << 4) + ((kqo & 1) << 3);
                    *reinterpret_cast<bf16x4*>(pT + se * ROWS * 64 + woff1) = PH[s];
                    *reinterpret_cast<bf16x4*>(pT + (SS + se) * ROWS * 64 + woff1) = PL[s];
                }
            }
        };
        // DIRECT: nothing is left in the value rows' region behind the edge barrier -- the images of slab 0 go there now
        if constexpr (DIRECT) {
            if (p.dw_partial && !(GML_ABL & 2)) { write_x(); write_slab(0); }
        }
        const bool late_issue = PINGPONG && (waveo & 4) != 0;
        if (!EARLYI && !late_issue) {
            latch();
            issue((GML_ABL & 1) ? g0 : min(g + 1, g1 - 1));
        }
        GML_T3(8);

        if (!DIRECT && p.dval && !(GML_ABL & 8)) {
            // dval += : a second launch over another slice of the input features (48-wide layers).  Not compiled into the ZINC shape
            // class (S = 8, 32 output columns: no config has 48-wide layers with 8 supports; the host refuses the flag there) -- the
            // mere presence of the branch cost that instantiation 1.5 % through its schedule (tools/ab.sh on one box, round 4)
            constexpr bool ACC = !(S == 8 && NOB == 2);
            if (ACC && (p.flags & GML_DVAL_ACCUM)) {
                for (int i = tid_o; i < ne * S; i += NT) p.dval[(int64_t)kb * S + i] += ea_l[i];
            } else if constexpr (VW > 1) {
                EV* dst = reinterpret_cast<EV*>(p.dval + (int64_t)kb * S);
                for (int i = tid_o; i < ne * (S / VW); i += NT) dst[i] = reinterpret_cast<const EV*>(ea_l)[i];
            } else {
                for (int i = tid_o; i < ne * S; i += NT) p.dval[(int64_t)kb * S + i] = ea_l[i];
            }
        }

        GML_T3(9);
        // ---- dX^T = W P^T: A[i = f][k = o] = W_s[f][o] comes transposed out of the [s][o][f] image: lane (t, kq) passes the
        //      address of row o = 8 kq + 4 h + (t >> 2), 8-byte chunk 4 fb + (t & 3), and receives W[8 kq + 4 h + j][16 fb + t]
        if ((DZ || !HAD) && p.dx && !(GML_ABL & 32)) {
            if constexpr (DZ) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) a += dzv[q] * *reinterpret_cast<const f32x4*>(wm_l + q * 32 + 16 * fb + 4 * kqo);
                    dxa[fb] = a;
                }
            } else if (!(p.flags & GML_ACCUM)) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const int tj = r16o >> 2, tc = r16o & 3;
            int aoff[2][NFB];                                // byte offsets inside one support's image
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    // (NOB = 1: k slots 8 kq + j, j < 4, are outputs 4 kq + j; the slots j >= 4 meet zeros in P -- any finite A will do)
                    const int o = NOB == 2 ? 8 * kqo + 4 * h + tj : 4 * kqo + tj, cidx = 4 * (fb & 1) + tc;
                    aoff[h][fb] = (fb >> 1) * (C::W_IMG * 2) + o * 64 + ((((cidx >> 1) ^ gml_wkey3(o)) & 3) << 4) + ((cidx & 1) << 3);
                }
            const unsigned char* Wh8 = reinterpret_cast<const unsigned char*>(W_h);
            const unsigned char* Wl8 = reinterpret_cast<const unsigned char*>(W_l);
            bf16x8 vh[2][NFB], vl[2][NFB];
            auto fragx = [&](int s, int st) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    vh[st][fb] = gml_tr_frag(Wh8 + s * (WO * 64) + aoff[0][fb], Wh8 + s * (WO * 64) + aoff[1][fb]);
                    vl[st][fb] = gml_tr_frag(Wl8 + s * (WO * 64) + aoff[0][fb], Wl8 + s * (WO * 64) + aoff[1][fb]);
                }
            };
            fragx(0, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int st = s & 1;
                if (s + 1 < S) fragx(s + 1, st ^ 1);
                bf16x8 ph, pl;
                if constexpr (NOB == 2) { ph = PH[s]; pl = PL[s]; }
                else {
                    const __bf16 z = (__bf16)0.f;
                    ph = bf16x8{PH[s][0], PH[s][1], PH[s][2], PH[s][3], z, z, z, z};
                    pl = bf16x8{PL[s][0], PL[s][1], PL[s][2], PL[s][3], z, z, z, z};
                }
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
            if constexpr (DZ) {
                if (p.relu_cols > 0) {                       // features 16 fb + 4 kq + reg of the own row: their mask bits sit in the lane
#pragma unroll                                               // (r16, 2 fb + (kq >> 1)), nibble kq & 1
                    for (int fb = 0; fb < NFB; ++fb) {
                        const unsigned mm = (unsigned)__shfl((int)xpos, r16o + 16 * (2 * fb + (kqo >> 1))) >> (4 * (kqo & 1));
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            if (16 * fb + 4 * kqo + reg < p.relu_cols && !((mm >> reg) & 1u)) dxa[fb][reg] = 0.f;
                    }
                }
            }
            float* dr = p.dx + (r0 + row) * p.lddx + 4 * kqo;
            if (GML_ABL & 8) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) asm volatile("" :: "v"(dxa[fb]));
            } else if (dxv) {
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

        if (late_issue) {
            latch();
            issue((GML_ABL & 1) ? g0 : min(g + 1, g1 - 1));
        }
        GML_T3(5);
        // ---- dW += X^T P over the rows of the group.  Row-major bf16 images [position = wave*16 + r16][32 channels]
        //      (a lane's 8 channels = one 16-byte chunk, XOR gml_tkey3(position)); the contraction reads them transposed.
        if (p.dw_partial && !(GML_ABL & 2)) {
            if constexpr (!DIRECT) write_x();
            // transposing-read offsets of K step 0: lane (t, kq): position 8 kq + 4 h + (t >> 2), chunk 4 blk + (t & 3)
            const int tj = r16o >> 2, tc = r16o & 3;
            auto roffs = [&](int h, int blk) {               // [h][16-wide channel block of a 32-channel row]
                const int ps = 8 * kqo + 4 * h + tj, cidx = 4 * blk + tc;
                return ps * 64 + ((((cidx >> 1) ^ gml_tkey3(ps)) & 3) << 4) + ((cidx & 1) << 3);
            };
            // wave's blocks b = wave + NW i: ob = b % NOB (the same for every i: NW is even), fb = (b / NOB) % NFB, se = (b / NOB) / NFB.
            // ASHARE (NW a multiple of NOB NFB): fb is the same for every i as well -- one X fragment serves the wave's blocks.
            constexpr int NB_ = (C::NBLK >= NW) ? BPW : 1;         // (fewer blocks than waves: one block on the first waves)
            constexpr int NA_ = C::ASHARE ? 1 : NB_;               // X fragments per K step
            const int ob = wave % NOB;
            const int offB[2] = {roffs(0, ob), roffs(1, ob)};
            int offA[NA_][2], seo[NB_];
#pragma unroll
            for (int i = 0; i < NB_; ++i) {
                const int b = min(wave + NW * i, C::NBLK - 1);           // (a wave without an i-th block re-reads its last one)
                seo[i] = ((b / NOB) / NFB) * ROWS * 64;
                if (i < NA_) {
                    const int fb = (b / NOB) % NFB;
                    offA[i][0] = (fb >> 1) * C::XT_IMG + roffs(0, fb & 1);
                    offA[i][1] = (fb >> 1) * C::XT_IMG + roffs(1, fb & 1);
                }
            }
#pragma unroll
            for (int sl = 0; sl < C::NSLAB; ++sl) {
                if (!(DIRECT && sl == 0)) {
                    __syncthreads();                         // slab region free: the dval copy-out (sl == 0) or the
                                                             // previous slab's fragment reads are done in every wave
                    write_slab(sl);
                }
                if constexpr (LATEC) {
                    if (sl == C::NSLAB - 1) commit(gi_c, (int)min((int64_t)ROWS, p.nrows - (int64_t)min(g + 1, g1 - 1) * ROWS));
                }
                __syncthreads();
                GML_T3(11);
                // The fragments of K step st + 1 are requested before the MFMAs of step st, and the blocks' accumulator
                // chains are interleaved (one chain of dependent MFMAs behind its own reads is pure latency).
                if (wave < C::NBLK) {
                    bf16x8 fah[2][NA_], fal[2][NA_], fbh[2][NB_], fbl[2][NB_];
                    auto fragw = [&](int st, int sg) {
                        const unsigned char* xa = xT + st * 2048;
#pragma unroll
                        for (int i = 0; i < NA_; ++i) {
                            fah[sg][i] = gml_tr_frag(xa + offA[i][0], xa + offA[i][1]);
                            fal[sg][i] = gml_tr_frag(xa + ROWS * 64 + offA[i][0], xa + ROWS * 64 + offA[i][1]);
                        }
#pragma unroll
                        for (int i = 0; i < NB_; ++i) {
                            const unsigned char* pa = pT + seo[i] + st * 2048;
                            fbh[sg][i] = gml_tr_frag(pa + offB[0], pa + offB[1]);
                            fbl[sg][i] = gml_tr_frag(pa + SS * ROWS * 64 + offB[0], pa + SS * ROWS * 64 + offB[1]);
                        }
                    };
                    f32x4 d[NB_];
#pragma unroll
                    for (int i = 0; i < NB_; ++i) d[i] = dwacc[sl][i];
                    fragw(0, 0);
#pragma unroll
                    for (int st = 0; st < ROWS / 32; ++st) {
                        const int sg = st & 1;
                        if (st + 1 < ROWS / 32) fragw(st + 1, sg ^ 1);
#pragma unroll
                        for (int i = 0; i < NB_; ++i) d[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[sg][i < NA_ ? i : 0], fbh[sg][i], d[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NB_; ++i) d[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[sg][i < NA_ ? i : 0], fbl[sg][i], d[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NB_; ++i) d[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[sg][i < NA_ ? i : 0], fbh[sg][i], d[i], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 4 * NA_ + 4 * NB_, 2);
#pragma unroll
                    for (int st = 0; st < ROWS / 32; ++st) {
                        if (st + 1 < ROWS / 32) __builtin_amdgcn_sched_group_barrier(0x100, 4 * NA_ + 4 * NB_, 2);
                        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NB_, 2);
                    }
#pragma unroll
                    for (int i = 0; i < NB_; ++i)
                        if (wave + NW * i < C::NBLK) dwacc[sl][i] = d[i];
                }
                GML_T3(7);
            }
        } else if constexpr (LATEC) {
            commit(gi_c, (int)min((int64_t)ROWS, p.nrows - (int64_t)min(g + 1, g1 - 1) * ROWS));
            __syncthreads();
        }
    }

    GML_T3(6);
#ifdef GML_BWD2_TIMING
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) atomicAdd(&p.prof[i], (unsigned long long)tacc_[i]);
    }
#endif
    // ---- one dW partial per workgroup: block b of slab sl: D[i = f][j = o], lane (o = r16, kq): f = 16 fb + 4 kq + reg
    if (p.dw_partial && g0 < g1) {
        float* out = p.dw_partial + (int64_t)wg * S * p.Fin * p.Fout;
#pragma unroll
        for (int sl = 0; sl < C::NSLAB; ++sl)
#pragma unroll
            for (int i = 0; i < BPW; ++i) {
                const int b = wave + NW * i;
                if (b < C::NBLK) {
                    const int ob = b % NOB, fb = (b / NOB) % NFB, s = sl * SS + (b / NOB) / NFB;
                    const int o = ob * 16 + r16;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int f = fb * 16 + 4 * kq + reg;
                        if (f < p.Fin && o < p.Fout) out[((int64_t)s * p.Fin + f) * p.Fout + o] = dwacc[sl][i][reg];
                        if constexpr (HAD) {                 // dW_s[f][30 + u], s < 2: row 2 s + u of [dw11; dw12]
                            if (f < p.Fin && o >= 30 && s < 2)
                                p.hpart[(int64_t)wg * (4 * p.Fin + 4 + p.Fout) + (2 * s + (o - 30)) * p.Fin + f] = dwacc[sl][i][reg];
                        }
                    }
                }
            }
    }
    if constexpr (HAD) {
        __syncthreads();                                     // every wave's record is complete
        if (g0 < g1 && tid < 36) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += bsum[w * 36 + tid];
            float* hp = p.hpart + (int64_t)wg * (4 * p.Fin + 4 + p.Fout) + 4 * p.Fin;
            if (tid >= 32) hp[tid - 32] = v;                 // db11 | db12
            else if (tid < p.Fout) hp[4 + tid] = v;          // dcb
        }
    }
}

template <int S, int NFB, int NW, int NOB = 2>
int gml_launch_bwd3(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st);

// the DZ form (dx = conv part + dz . wmix) is compiled for the shape class that uses it: ZINC's layers (S = 8, Fin <= 32,
// 2 x nout2 = 4 Hadamard columns); float4-addressable x / dx rows
template <int S, int NFB, int NW, bool EN>
struct GmlBwd3Dz {
    static int go(const GmlBwdParams&, dim3, size_t, hipStream_t) { return GML_E_UNSUPPORTED; }
};
template <int S, int NFB, int NW>
struct GmlBwd3Dz<S, NFB, NW, true> {
    static int go(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st) {
        if (!p.xvec || !p.dxvec || p.nmix < 1 || p.nmix > 4 || !p.wmix || !p.dx) return GML_E_UNSUPPORTED;
        GML_ALLOW_BIG_LDS(rc, (&gml_k_spectconv_bwd3<S, NFB, NW, true, true>), 160 * 1024)
        if (rc != hipSuccess) return (int)rc;
        hipLaunchKernelGGL((gml_k_spectconv_bwd3<S, NFB, NW, true, true>), grid, dim3(64 * NW), lds, st, p);   /* (lds includes the 512 bytes) */
        return gml_launch_status();
    }
};
#define GML_BWD3_HAS_DZ(SV, NFBV, NWV) ((SV) == 8 && (NFBV) == 2 && (NWV) == 8)

// the HAD form (gml_bwd3_fam_h.hip): lds = the plan's bytes + GML_BWD3_HAD_LDS
int gml_launch_bwd3_had(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st);

#define GML_DEFINE_BWD3(SV, NFBV, NWV)                                                                       \
    template <>                                                                                              \
    int gml_launch_bwd3<SV, NFBV, NWV>(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st) {       \
        static_assert(GmlBwd3Cfg<SV, NFBV, NWV>::OK, "unsupported shape");                                   \
        if (p.dz != nullptr) return GmlBwd3Dz<SV, NFBV, NWV, GML_BWD3_HAS_DZ(SV, NFBV, NWV)>::go(p, grid, lds, st);  \
        GML_ALLOW_BIG_LDS(rc1, (&gml_k_spectconv_bwd3<SV, NFBV, NWV, true>), 160 * 1024)                     \
        GML_ALLOW_BIG_LDS(rc0, (&gml_k_spectconv_bwd3<SV, NFBV, NWV, false>), 160 * 1024)                    \
        if (rc1 != hipSuccess) return (int)rc1;                                                              \
        if (rc0 != hipSuccess) return (int)rc0;                                                              \
        if (p.xvec) hipLaunchKernelGGL((gml_k_spectconv_bwd3<SV, NFBV, NWV, true>), grid, dim3(64 * NWV), lds, st, p);  \
        else hipLaunchKernelGGL((gml_k_spectconv_bwd3<SV, NFBV, NWV, false>), grid, dim3(64 * NWV), lds, st, p);        \
        return gml_launch_status();                                                                          \
    }

// NOB = 1 (Fout <= 16): no dz hand-over form
#define GML_DEFINE_BWD3_N1(SV, NFBV, NWV)                                                                    \
    template <>                                                                                              \
    int gml_launch_bwd3<SV, NFBV, NWV, 1>(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st) {    \
        static_assert(GmlBwd3Cfg<SV, NFBV, NWV, 1>::OK, "unsupported shape");                                \
        if (p.dz != nullptr) return GML_E_UNSUPPORTED;                                                       \
        GML_ALLOW_BIG_LDS(rc1, (&gml_k_spectconv_bwd3<SV, NFBV, NWV, true, false, 1>), 160 * 1024)           \
        GML_ALLOW_BIG_LDS(rc0, (&gml_k_spectconv_bwd3<SV, NFBV, NWV, false, false, 1>), 160 * 1024)          \
        if (rc1 != hipSuccess) return (int)rc1;                                                              \
        if (rc0 != hipSuccess) return (int)rc0;                                                              \
        if (p.xvec) hipLaunchKernelGGL((gml_k_spectconv_bwd3<SV, NFBV, NWV, true, false, 1>), grid, dim3(64 * NWV), lds, st, p);  \
        else hipLaunchKernelGGL((gml_k_spectconv_bwd3<SV, NFBV, NWV, false, false, 1>), grid, dim3(64 * NWV), lds, st, p);        \
        return gml_launch_status();                                                                          \
    }
