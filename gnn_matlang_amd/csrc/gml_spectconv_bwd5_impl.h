// Fused backward, bf16x3, 12-wave form for the ZINC shape class (S = 8 supports, 16 < Fin <= 32, 16 < Fout <= 32): same outputs as
// gml_k_spectconv_bwd3 (gml_spectconv_bwd3_impl.h), bit for bit
//
//   dX = sum_s A_s (G W_s^T),   dval[e,s] = < X[src] W_s, G[dst] >,   dW_s = X^T (A_s G)
//
// bwd3 keeps Z and P of eight supports -- 128 accumulators -- beside everything else in 256 VGPRs: two waves per
// SIMD, and every wave walks every phase (prefetch issue, commit, projections, edge loop, image writes, row contraction) in lockstep.
// Round 5's measurements (profiles/r05_bwd3_variants_ab.txt) say what that costs and what does not help: two independent
// half-size workgroups per CU, in phase or a half group out of phase, run no faster -- two waves per SIMD do not hide their own
// dependent chains whatever the partner does.  This form cuts the accumulator set instead:
//   * the edge loop runs TWICE over a tile, once with Z live (d = <Z, G[dst]> -> dval, straight to global memory) and once with P
//     live (P += val G[dst]; value rows straight from global memory): 64 accumulators at a time, <= 168 VGPRs, three waves per SIMD;
//   * the third wave of every SIMD is a HELPER wave: the four helpers prefetch and commit the next group's column ids / G window /
//     row pointers and run the whole row contraction dW += X^T P out of the LDS images (the 8 x 4 output blocks' accumulators live in
//     THEIR registers), while the eight compute waves go on with the next phase.  Compute waves never issue a prefetch, never
//     commit, never contract.
// Schedule (four workgroup barriers per group): the helpers' row contractions -- matrix pipe -- sit beside the compute waves'
// VALU-bound edge passes, never beside their projections (the first version of this kernel paired Z with one contraction and dX
// with the other: every matrix instruction of the SIMD in the same two windows, +11 % against bwd3):
//        compute waves                                              | helper waves
//   [1]  edge pass P of group g, split P                            | contraction of slab 1 of group g - 1
//   [2]  dX projection + dx store, write X image + P slab 0         | commit g + 1 (its regions are dead: both passes of g are done)
//   [3]  Z projection + edge pass Z of group g + 1                  | contraction of slab 0 of group g
//   [4]  write P slab 1                                             | prefetch g + 2 into registers
// LDS as bwd3's VALG layout: W image 32 KB, row pointers, column ids, G window, X image 16 KB, one P slab 64 KB (nothing aliased).
#pragma once
#include "gml_spectconv_bwd3_impl.h"

template <int NFB_>
struct GmlBwd5Cfg {
    static constexpr int S = 8, NOB = 2, NFB = NFB_, ROWS = 128, NTC = 512, NTH = 256, NT = NTC + NTH;
    static constexpr int LDG = 36, W_HALF = S * 32 * 32, W_BYTES = 4 * W_HALF, XT_BYTES = 2 * ROWS * 64, PT_BYTES = 2 * 4 * ROWS * 64;
    static constexpr int ECAP_MAX = 1024, XCAP_MAX = 224, GREC = 4 + ROWS / 4;
    __host__ __device__ static size_t lds_bytes(int ecap, int xcap) {
        return (size_t)W_BYTES + (ROWS + 8) * 4 + (size_t)ecap * 4 + (size_t)xcap * LDG * 4 + XT_BYTES + PT_BYTES + 512;
    }
};

template <int NFB, bool XV, bool DZ>
__global__ __launch_bounds__(768, 1) void gml_k_spectconv_bwd5(const GmlBwdParams p) {
    using C = GmlBwd5Cfg<NFB>;
    static_assert(NFB == 2, "compiled for 16 < Fin <= 32");
    constexpr int S = 8, NOB = 2, NH = 4, GC = 8, LDG = C::LDG, ROWS = C::ROWS, WO = 32, SS = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* W_h = reinterpret_cast<__bf16*>(lds_raw);        // [s][o][f], chunks XOR gml_wkey3(o)
    __bf16* W_l = W_h + C::W_HALF;
    int* rp_l = reinterpret_cast<int*>(lds_raw + C::W_BYTES);
    int* col_l = rp_l + ROWS + 8;
    float* gs = reinterpret_cast<float*>(col_l + p.ecap);
    unsigned char* xT = reinterpret_cast<unsigned char*>(gs + (size_t)p.xcap * LDG);
    unsigned char* pT = xT + C::XT_BYTES;
    float* wm_l = reinterpret_cast<float*>(pT + C::PT_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = wg * p.groups_per_wg;
    const int g1 = min(g0 + p.groups_per_wg, p.ngroups);
    const bool helper = wave >= 8;

    if constexpr (DZ) {
        if (tid < 128) wm_l[tid] = ((tid >> 5) < p.nmix && (tid & 31) < p.Fin) ? ((tid >> 5) < p.nmix1 ? p.wmix[(tid >> 5) * p.Fin + (tid & 31)] : p.wmix2[((tid >> 5) - p.nmix1) * p.Fin + (tid & 31)]) : 0.f;
    }
    for (int e = tid; e < S * WO * 32; e += C::NT) {          // W -> bf16 (hi, lo) image, zero padded to 32 x 32
        const int f = e & 31, o = (e >> 5) % WO, s = (e >> 5) / WO;
        const float v = (f < p.Fin && o < p.Fout) ? p.w[((int64_t)s * p.Fin + f) * p.Fout + o] : 0.f;
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        const int i = (s * WO + o) * 32 + ((((f >> 3) ^ gml_wkey3(o)) & 3) << 3) + (f & 7);
        W_h[i] = h; W_l[i] = l;
    }
    if (g0 >= g1) return;                                     // (uniform: no barrier below is reached by anyone)
    const int etot = p.rowptr[p.nrows];
    auto rsrc = [](const void* base, int64_t nbytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(nbytes < 0 ? 0 : (nbytes > 0x7fffff00 ? 0x7fffff00 : nbytes)), 0x00020000);
    };
    // {first edge, #edges, first window row, window rows} of a group: loaded one group ahead (record_raw), made wave-uniform at its use
    auto record_raw = [&](int g) -> int4 { return *reinterpret_cast<const int4*>(p.ginfo + (int64_t)g * C::GREC); };
    auto uniform4 = [](const int4 v) -> int4 {
        return int4{__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z),
                    __builtin_amdgcn_readfirstlane(v.w)};
    };

    if (helper) {
        // =========================================================================================================== helper waves
        const int ht = tid - C::NTC, hw = wave - 8;            // 0..255, 0..3
        constexpr int NC = C::ECAP_MAX / C::NTH, NG4 = (C::XCAP_MAX * GC + C::NTH - 1) / C::NTH;     // 4, 7
        int cv[NC], rpv = 0;
        f32x4 gv4[NG4];
        const int voff_g = (ht / GC) * (int)p.ldg * 4 + (ht % GC) * 16;
        const int ldgb = (int)p.ldg * 4;
        auto vec_group = [&](const int4 gi) { return p.gvec && gi.y <= C::ECAP_MAX && gi.w <= C::XCAP_MAX; };
        auto issue = [&](int g, const int4 gi) {               // the group's row pointers, column ids and G window into registers
            const int64_t r0 = (int64_t)g * ROWS;
            const auto rs_rp = rsrc(p.rowptr + r0, (p.nrows + 1 - r0) * 4);
            const auto rs_col = rsrc(p.col + gi.x, ((int64_t)etot - gi.x) * 4);
            const auto rs_g = rsrc(p.g + (int64_t)gi.z * p.ldg, (p.nrows - gi.z) * p.ldg * 4);
            rpv = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_rp, ht * 4, 0, 0);
#pragma unroll
            for (int t = 0; t < NC; ++t) cv[t] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_col, ht * 4, C::NTH * 4 * t, 0);
#pragma unroll
            for (int t = 0; t < NG4; ++t) gv4[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, voff_g, t * (C::NTH / GC) * ldgb, 0));
        };
        auto commit = [&](int g, const int4 gi) {
            const int kb = gi.x, ne = gi.y, lo = gi.z, nwin = gi.w;
            const int nr = (int)min((int64_t)ROWS, p.nrows - (int64_t)g * ROWS);
            if (ht <= nr) rp_l[ht] = rpv;
            if (vec_group(gi)) {
#pragma unroll
                for (int t = 0; t < NC; ++t) { const int i = ht + C::NTH * t; if (i < ne) col_l[i] = cv[t] - lo; }
#pragma unroll
                for (int t = 0; t < NG4; ++t) {
                    const int i = ht + C::NTH * t;
                    if (i < nwin * GC)
                        *reinterpret_cast<f32x4*>(gs + (i / GC) * LDG + (i % GC) * 4) = ((i % GC) * 4 < p.Fout) ? gv4[t] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            } else {                                           // a group beyond the register-batched bounds: rolled loops
                for (int i = ht; i < ne; i += C::NTH) col_l[i] = p.col[kb + i] - lo;
                for (int i = ht; i < nwin * 32; i += C::NTH) {
                    const int rr = i / 32, o = i % 32;
                    gs[rr * LDG + o] = (o < p.Fout) ? p.g[(int64_t)(lo + rr) * p.ldg + o] : 0.f;
                }
            }
        };
        // row contraction: this wave owns the output blocks (fb, ob) = (hw >> 1, hw & 1) of all eight supports
        const int ob = hw & 1, fb = hw >> 1;
        f32x4 dwacc[2][SS];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int se = 0; se < SS; ++se) dwacc[sl][se] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int tj = r16 >> 2, tc = r16 & 3;
        int roff[2][2];                                        // [h][16-wide channel block]: transposing-read offsets of K step 0
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int ps = 8 * kq + 4 * h + tj, cidx = 4 * blk + tc;
                roff[h][blk] = ps * 64 + ((((cidx >> 1) ^ gml_tkey3(ps)) & 3) << 4) + ((cidx & 1) << 3);
            }
        auto contract = [&](int sl) {                          // dW[4 sl + se] += X^T P over the 128 rows of the images
#pragma unroll
            for (int st = 0; st < ROWS / 32; ++st) {
                const unsigned char* xa = xT + st * 2048;
                const bf16x8 fah = gml_tr_frag(xa + roff[0][fb], xa + roff[1][fb]);
                const bf16x8 fal = gml_tr_frag(xa + ROWS * 64 + roff[0][fb], xa + ROWS * 64 + roff[1][fb]);
                bf16x8 fbh[SS], fbl[SS];
#pragma unroll
                for (int se = 0; se < SS; ++se) {
                    const unsigned char* pa = pT + se * ROWS * 64 + st * 2048;
                    fbh[se] = gml_tr_frag(pa + roff[0][ob], pa + roff[1][ob]);
                    fbl[se] = gml_tr_frag(pa + SS * ROWS * 64 + roff[0][ob], pa + SS * ROWS * 64 + roff[1][ob]);
                }
#pragma unroll
                for (int se = 0; se < SS; ++se) dwacc[sl][se] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal, fbh[se], dwacc[sl][se], 0, 0, 0);
#pragma unroll
                for (int se = 0; se < SS; ++se) dwacc[sl][se] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah, fbl[se], dwacc[sl][se], 0, 0, 0);
#pragma unroll
                for (int se = 0; se < SS; ++se) dwacc[sl][se] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah, fbh[se], dwacc[sl][se], 0, 0, 0);
            }
        };
        const bool want_dw = p.dw_partial != nullptr;
        int4 gi_n = uniform4(record_raw(g0));
        issue(g0, gi_n);
        commit(g0, gi_n);
        __syncthreads();                                       // [0] W image, wmix rows, the first group's commit
        gi_n = uniform4(record_raw(min(g0 + 1, g1 - 1)));
        issue(min(g0 + 1, g1 - 1), gi_n);                      // (the compute waves: Z projection + pass Z of the first group)
        int4 raw_n = record_raw(min(g0 + 2, g1 - 1));
        __syncthreads();                                       // [4] of "group g0 - 1"
        for (int g = g0; g < g1; ++g) {
            if (want_dw && g > g0) contract(1);                // [1] slab 1 of the previous group, beside the compute waves' pass P
            __syncthreads();
            commit(min(g + 1, g1 - 1), gi_n);                  // [2] (registers loaded in [4] of the previous iteration)
            __syncthreads();
            if (want_dw) contract(0);                          // [3] beside the compute waves' pass Z of the next group
            __syncthreads();
            gi_n = uniform4(raw_n);                            // [4] prefetch of the group after next
            raw_n = record_raw(min(g + 3, g1 - 1));
            issue(min(g + 2, g1 - 1), gi_n);
            __syncthreads();
        }
        if (want_dw) {
            contract(1);                                       // slab 1 of the last group
            // one dW partial per workgroup: D[i = f][j = o], lane (o = r16, kq): f = 16 fb + 4 kq + reg
            float* out = p.dw_partial + (int64_t)wg * S * p.Fin * p.Fout;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int se = 0; se < SS; ++se) {
                    const int s = sl * SS + se, o = ob * 16 + r16;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int f = fb * 16 + 4 * kq + reg;
                        if (f < p.Fin && o < p.Fout) out[((int64_t)s * p.Fin + f) * p.Fout + o] = dwacc[sl][se][reg];
                    }
                }
        }
        return;
    }

    // =============================================================================================================== compute waves
    const auto rs_xall = rsrc(p.x, p.nrows * p.ldx * 4 > 0x7fffff00 ? 0 : p.nrows * p.ldx * 4);   // (XV rows through 32-bit offsets when they fit)
    const bool x32 = p.nrows * p.ldx * 4 <= 0x7fffff00;
    auto row_of = [&](int g) -> int { return reinterpret_cast<const unsigned char*>(p.ginfo + (int64_t)g * C::GREC + 4)[wave * 16 + r16]; };
    float xb[8];
    auto load_x = [&](int g, int row) {                        // the lane's own x row, features 8 kq .. 8 kq + 7 (clamped, masked later)
        const int64_t rr = min((int64_t)g * ROWS + row, p.nrows - 1);
        if constexpr (XV) {
            const int f4max = (p.Fin + 3) / 4 * 4 - 4;
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                f32x4 t;
                if (x32) t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_xall, (int)(rr * p.ldx + min(8 * kq + 4 * q4, f4max)) * 4, 0, 0));
                else t = *reinterpret_cast<const f32x4*>(p.x + rr * p.ldx + min(8 * kq + 4 * q4, f4max));
                xb[4 * q4] = t.x; xb[4 * q4 + 1] = t.y; xb[4 * q4 + 2] = t.z; xb[4 * q4 + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) xb[t] = p.x[rr * p.ldx + min(8 * kq + t, p.Fin - 1)];
        }
    };
    // per-group state of a lane: the group whose pass P / projections run in [1], [2] ("cur") and the one whose pass Z runs in [3]
    struct Grp { int kb, ne, row, kbeg, kend; bool rvalid; int64_t r0; bf16x8 xh, xl; unsigned xpos; };
    auto ldg_row = [&](int dstl, f32x2 (&gv)[NH]) {            // the lane's 8 columns of the destination's G row
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(gs + dstl * LDG + 8 * kq + 4 * ob);
            gv[2 * ob] = f32x2{t0.x, t0.y}; gv[2 * ob + 1] = f32x2{t0.z, t0.w};
        }
    };
    // ---- Z projection + edge pass Z of group gz (its commit is visible): fills the group's state, stores its dval rows
    auto zpass = [&](int gz, const int4 gi, int row, Grp& q) {
        q.kb = gi.x; q.ne = gi.y; q.row = row;
        q.r0 = (int64_t)gz * ROWS;
        const int nr = (int)min((int64_t)ROWS, p.nrows - q.r0);
        q.rvalid = row < nr;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (!(q.rvalid && 8 * kq + j < p.Fin)) xb[j] = 0.f;
        gml_split8(xb, q.xh, q.xl);
        q.xpos = 0;
        if constexpr (DZ) {
#pragma unroll
            for (int j = 0; j < 8; ++j) q.xpos |= (xb[j] > 0.f ? 1u : 0u) << j;
        }
        q.kbeg = q.rvalid ? rp_l[row] - q.kb : 0;
        q.kend = q.rvalid ? rp_l[row + 1] - q.kb : 0;
        const auto dvrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dval ? p.dval + (int64_t)q.kb * S : p.x), 0, p.dval ? q.ne * (S * 4) : 0, 0x00020000);
        // Z^T = W^T X^T: MFMA row i of block ob carries o = 8 (i >> 2) + 4 ob + (i & 3): lane kq receives its 8 consecutive outputs
        f32x2 Z[S][NH];
        const int oa0 = 8 * (r16 >> 2) + (r16 & 3);
        bf16x8 wh[2][NOB], wl[2][NOB];
        auto frag = [&](int s_, int st) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                const int oa = oa0 + 4 * ob;
                const int off = (s_ * WO + oa) * 32 + (((kq ^ gml_wkey3(oa)) & 3) << 3);
                wh[st][ob] = *reinterpret_cast<const bf16x8*>(W_h + off);
                wl[st][ob] = *reinterpret_cast<const bf16x8*>(W_l + off);
            }
        };
        frag(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const int st = s_ & 1;
            if (s_ + 1 < S) frag(s_ + 1, st ^ 1);
            f32x4 dd[NOB];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) dd[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) dd[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[st][ob], q.xh, dd[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) dd[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[st][ob], q.xl, dd[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) dd[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[st][ob], q.xh, dd[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) { Z[s_][2 * ob] = f32x2{dd[ob][0], dd[ob][1]}; Z[s_][2 * ob + 1] = f32x2{dd[ob][2], dd[ob][3]}; }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NOB, 0);
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            if (s_ + 1 < S) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NOB, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NOB, 0);
        }
        // software pipeline over the lane's edges (two register sets, no rotation): the G row of edge k + 1 and the column id of edge
        // k + 2 are requested before the arithmetic of edge k -- with 64 accumulators per pass there is room for it, and a pass of
        // 32 packed FMAs per edge no longer covers the loop's two dependent LDS round trips by itself
        auto edge_z = [&](int k, const f32x2 (&gv)[NH]) {
            float d[S];
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_) {
                f32x2 a2 = f32x2{0.f, 0.f};
#pragma unroll
                for (int h = 0; h < NH; ++h) a2 = Z[s_][h] * gv[h] + a2;
                d[s_] = a2.x + a2.y;
            }
            float tot2[2];                                     // fold slot j of chunk c = support 2 j + c: lane kq ends with supports 2 kq, 2 kq + 1
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const auto a01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[c]), __float_as_uint(d[2 + c]), false, false);
                const auto a23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[4 + c]), __float_as_uint(d[6 + c]), false, false);
                const float c01 = __uint_as_float(a01[0]) + __uint_as_float(a01[1]);
                const float c23 = __uint_as_float(a23[0]) + __uint_as_float(a23[1]);
                const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(c01), __float_as_uint(c23), false, false);
                tot2[c] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
            }
            typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(u32x2_{__float_as_uint(tot2[0]), __float_as_uint(tot2[1])}, dvrs, (k * S + 2 * kq) * 4, 0, 0);
        };
        int k = q.kbeg;
        if (k < q.kend) {
            const int klast = q.kend - 1;
            f32x2 gA[NH], gB[NH];
            ldg_row(col_l[k], gA);
            int cn = col_l[min(k + 1, klast)];
            for (;;) {
                const int c2 = col_l[min(k + 2, klast)];
                ldg_row(cn, gB);
                edge_z(k, gA);
                if (++k >= q.kend) break;
                cn = col_l[min(k + 2, klast)];
                ldg_row(c2, gA);
                edge_z(k, gB);
                if (++k >= q.kend) break;
            }
        }
    };

    int row_n = row_of(min(g0 + 1, g1 - 1));                   // rows travel two groups ahead of their use, records one
    int4 raw_n = record_raw(min(g0 + 1, g1 - 1));
    Grp cur;
    {
        const int row0 = row_of(g0);
        const int4 gi0 = uniform4(record_raw(g0));
        load_x(g0, row0);
        __syncthreads();                                       // [0] W image, wmix rows, the first group's commit
        zpass(g0, gi0, row0, cur);
        load_x(min(g0 + 1, g1 - 1), row_n);                    // the next group's x row: in flight across pass P
        __syncthreads();                                       // [4] of "group g0 - 1"
    }
    for (int g = g0; g < g1; ++g) {
        const int gn = min(g + 1, g1 - 1);
        const int64_t r0 = cur.r0;
        const int row = cur.row;
        const bool rvalid = cur.rvalid;
        // ---- [1] pass P of group g: P += val G[dst] (value rows from global memory: two rows in flight per lane), then P -> bf16 (hi, lo)
        bf16x8 PH[S], PL[S];
        {
            const auto vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.val + (int64_t)cur.kb * S), 0, cur.ne * (S * 4), 0x00020000);
            f32x2 P[S][NH];
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_)
#pragma unroll
                for (int h = 0; h < NH; ++h) asm volatile("v_pk_mov_b32 %0, 0, 0" : "=v"(P[s_][h]));
            auto ldval = [&](int k, float (&ev)[S]) {
                const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(vrs, k * (S * 4), 0, 0);
                const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(vrs, k * (S * 4) + 16, 0, 0);
                ev[0] = __uint_as_float(a.x); ev[1] = __uint_as_float(a.y); ev[2] = __uint_as_float(a.z); ev[3] = __uint_as_float(a.w);
                ev[4] = __uint_as_float(b.x); ev[5] = __uint_as_float(b.y); ev[6] = __uint_as_float(b.z); ev[7] = __uint_as_float(b.w);
            };
            auto edge_p = [&](const float (&ev)[S], const f32x2 (&gv)[NH]) {
#pragma unroll
                for (int s_ = 0; s_ < S; ++s_) {
                    const f32x2 e2 = f32x2{ev[s_], ev[s_]};
#pragma unroll
                    for (int h = 0; h < NH; ++h) P[s_][h] = e2 * gv[h] + P[s_][h];
                }
            };
            int k = cur.kbeg;
            if (k < cur.kend) {
                const int klast = cur.kend - 1;
                float eA[S], eB[S];
                f32x2 gA[NH], gB[NH];
                ldval(k, eA);
                ldg_row(col_l[k], gA);
                int cn = col_l[min(k + 1, klast)];
                for (;;) {
                    const int c2 = col_l[min(k + 2, klast)];
                    ldval(min(k + 1, klast), eB);
                    ldg_row(cn, gB);
                    edge_p(eA, gA);
                    if (++k >= cur.kend) break;
                    cn = col_l[min(k + 2, klast)];
                    ldval(min(k + 1, klast), eA);
                    ldg_row(c2, gA);
                    edge_p(eB, gB);
                    if (++k >= cur.kend) break;
                }
            }
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_) {
                const float pv[8] = {P[s_][0].x, P[s_][0].y, P[s_][1].x, P[s_][1].y, P[s_][2].x, P[s_][2].y, P[s_][3].x, P[s_][3].y};
                gml_split8(pv, PH[s_], PL[s_]);
            }
        }
        __syncthreads();                                       // [1] -> [2]: both passes of g are done in every wave; the images are free
        // ---- [2] dX^T = W P^T (A[i = f][k = o] comes transposed out of the [s][o][f] image, see bwd3), dx store, X image + slab 0
        const int pos = wave * 16 + r16;
        const int woff = pos * 64 + (((kq ^ gml_tkey3(pos)) & 3) << 4);
        auto write_slab = [&](int sl) {
#pragma unroll
            for (int se = 0; se < SS; ++se) {
                *reinterpret_cast<bf16x8*>(pT + se * ROWS * 64 + woff) = PH[sl * SS + se];
                *reinterpret_cast<bf16x8*>(pT + (SS + se) * ROWS * 64 + woff) = PL[sl * SS + se];
            }
        };
        if (p.dw_partial) {
            *reinterpret_cast<bf16x8*>(xT + woff) = cur.xh;
            *reinterpret_cast<bf16x8*>(xT + ROWS * 64 + woff) = cur.xl;
            write_slab(0);
        }
        if (p.dx) {
            f32x4 dxa[NFB];
            const bool dxv = p.dxvec != 0;
            if constexpr (DZ) {
                const f32x4 dzv = *reinterpret_cast<const f32x4*>(p.dz + min(r0 + row, p.nrows - 1) * 4);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) a += dzv[q4] * *reinterpret_cast<const f32x4*>(wm_l + q4 * 32 + 16 * fb + 4 * kq);
                    dxa[fb] = a;
                }
            } else if (p.flags & GML_ACCUM) {                  // old dx values: the lane's own row, features 16 fb + 4 kq .. + 3
                const float* dr = p.dx + min(r0 + row, p.nrows - 1) * p.lddx;
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
            } else {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const int tj = r16 >> 2, tc = r16 & 3;
            int aoff[2][NFB];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    const int o = 8 * kq + 4 * h + tj, cidx = 4 * fb + tc;
                    aoff[h][fb] = o * 64 + ((((cidx >> 1) ^ gml_wkey3(o)) & 3) << 4) + ((cidx & 1) << 3);
                }
            const unsigned char* Wh8 = reinterpret_cast<const unsigned char*>(W_h);
            const unsigned char* Wl8 = reinterpret_cast<const unsigned char*>(W_l);
            bf16x8 vh[2][NFB], vl[2][NFB];
            auto fragx = [&](int s_, int st) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    vh[st][fb] = gml_tr_frag(Wh8 + s_ * (WO * 64) + aoff[0][fb], Wh8 + s_ * (WO * 64) + aoff[1][fb]);
                    vl[st][fb] = gml_tr_frag(Wl8 + s_ * (WO * 64) + aoff[0][fb], Wl8 + s_ * (WO * 64) + aoff[1][fb]);
                }
            };
            fragx(0, 0);
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_) {
                const int st = s_ & 1;
                if (s_ + 1 < S) fragx(s_ + 1, st ^ 1);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl[st][fb], PH[s_], dxa[fb], 0, 0, 0);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[st][fb], PL[s_], dxa[fb], 0, 0, 0);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[st][fb], PH[s_], dxa[fb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * NFB, 1);
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_) {
                if (s_ + 1 < S) __builtin_amdgcn_sched_group_barrier(0x100, 4 * NFB, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NFB, 1);
            }
            if constexpr (DZ) {
                if (p.relu_cols > 0) {                         // features 16 fb + 4 kq + reg of the own row: their mask bits sit in lane (r16, 2 fb + (kq >> 1))
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb) {
                        const unsigned mm = (unsigned)__shfl((int)cur.xpos, r16 + 16 * (2 * fb + (kq >> 1))) >> (4 * (kq & 1));
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            if (16 * fb + 4 * kq + reg < p.relu_cols && !((mm >> reg) & 1u)) dxa[fb][reg] = 0.f;
                    }
                }
            }
            float* dr = p.dx + (r0 + row) * p.lddx + 4 * kq;
            if (dxv) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
                    if (rvalid && 16 * fb + 4 * kq < p.Fin) *reinterpret_cast<f32x4*>(dr + 16 * fb) = dxa[fb];
            } else {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        if (rvalid && 16 * fb + 4 * kq + reg < p.Fin) dr[16 * fb + reg] = dxa[fb][reg];
            }
        }
        __syncthreads();                                       // [2] -> [3]: slab 0 + X image visible; the next group's commit visible
        // ---- [3] Z projection + pass Z of the next group (beside the helpers' contraction of slab 0)
        if (g + 1 < g1) {
            const int4 gin = uniform4(raw_n);
            const int rown = row_n;
            raw_n = record_raw(min(g + 2, g1 - 1));
            row_n = row_of(min(g + 2, g1 - 1));
            zpass(g + 1, gin, rown, cur);                      // (PH / PL of slab 1 stay live across it: 32 registers beside the 64 of Z)
            load_x(min(g + 2, g1 - 1), row_n);
        }
        __syncthreads();                                       // [3] -> [4]: slab 0 consumed
        if (p.dw_partial) write_slab(1);
        __syncthreads();                                       // [4] -> [1]
    }
}

template <int NFB>
int gml_launch_bwd5(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st);

#define GML_BWD5_GO(NFBV, XVV, DZV)                                                                           \
    {                                                                                                        \
        GML_ALLOW_BIG_LDS(rc_, (&gml_k_spectconv_bwd5<NFBV, XVV, DZV>), 160 * 1024)                          \
        if (rc_ != hipSuccess) return (int)rc_;                                                              \
        hipLaunchKernelGGL((gml_k_spectconv_bwd5<NFBV, XVV, DZV>), grid, dim3(768), lds, st, p);            \
        return gml_launch_status();                                                                          \
    }
#define GML_DEFINE_BWD5(NFBV)                                                                                \
    template <>                                                                                              \
    int gml_launch_bwd5<NFBV>(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st) {                \
        if (p.dz != nullptr) {                                                                               \
            if (!p.xvec || !p.dxvec || p.nmix < 1 || p.nmix > 4 || !p.wmix || !p.dx) return GML_E_UNSUPPORTED; \
            GML_BWD5_GO(NFBV, true, true)                                                                    \
        }                                                                                                    \
        if (p.xvec) GML_BWD5_GO(NFBV, true, false)                                                           \
        GML_BWD5_GO(NFBV, false, false)                                                                      \
    }
