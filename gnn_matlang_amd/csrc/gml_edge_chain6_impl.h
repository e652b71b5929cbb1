// ML3Layer edge branch FORWARD with fp32-class products on the bf16 matrix cores ("bf16x6"), S = Sout <= 8 (reference:
// /root/reference/libs/spect_conv.py:190-194, 205-207).  Same function and machine mapping as gml_edge_chain_impl.h (a wave works on
// tiles of 16 edges, the edge is the COLUMN of every 16x16 tile, D registers of one MFMA are the B operand of the next):
//
//   out = relu( W4 . [ relu(W1 e) ; tanh(W2 e) * tanh(W3 e) ] )
//
// Why it exists (round 6, profiles/r06_precision_diag.jsonl): with the two-piece chain (x = hi + lo, residual 2^-17, the raw supports
// even 2^-16) the learned supports carry ~5e-7 rms -- and after training that error, not the conv kernels' and not any backward kernel's,
// is what moves parameter gradients to 1e-3 .. 1e-2 of their term sums (nearly dead relu units downstream decide differently).  With
// this branch in exact arithmetic the same gradients sit at 3e-5.  Here every fp32 operand is cut into THREE bf16 pieces
// x = h + m + l (8 + 8 + 8 significant bits: exact) and the products down to 2^-24 are kept:
//
//   layer 1 (K = S <= 8 per product): two K = 32 instructions per weight matrix; lane group g carries one (weight piece, support
//            piece) product each:  A: (h,h) (h,m) (m,h) (m,m)     B: (h,l) (l,h) (m,l) (l,m)       -- everything but (l,l)
//   layer 2 (K = 4 S = 32): six instructions  (h,h) (h,m) (m,h) (h,l) (m,m) (l,h)
//
// The supports are read as fp32 rows (32 bytes per edge, what the pre-split rows of the two-piece chain cost) and cut in registers.
// The kernel is bound by its loads and stores more than by issue, so the doubled MFMA count costs little (DESIGN s4.3).
// The BACKWARD kernel is unchanged: it recomputes its intermediates with the two-piece chain -- backward arithmetic was measured
// not to matter (same file) -- so an output within ~1e-6 of zero may be masked differently by the two; the value there is ~0.
#pragma once
#include "gml_edge_chain_impl.h"

struct GmlOp3 { bf16x8 h, m, l; };

// max(x, 0) as ONE instruction: non-negative floats order like their bit patterns and every negative one has the sign bit, so the signed
// integer maximum with 0 is the float's (fmaxf costs two: the compiler quiets a possible signalling NaN with v_max x, x first)
__device__ __forceinline__ float gml_relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// (x0, x1) -> packed bf16 pairs h, m, l with h + m + l = x (round to nearest three times; the last residual is exact)
__device__ __forceinline__ void gml_split3_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = gml_pack2(x0, x1);
    const float r0 = x0 - gml_bf_lo(h), r1 = x1 - gml_bf_hi(h);
    m = gml_pack2(r0, r1);
    l = gml_pack2(r0 - gml_bf_lo(m), r1 - gml_bf_hi(m));
}
__device__ __forceinline__ GmlOp3 gml_wop3(const float (&v)[8]) {
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gml_split3_pair(v[2 * j], v[2 * j + 1], h[j], m[j], l[j]);
    GmlOp3 o;
    o.h = gml_op(h[0], h[1], h[2], h[3]);
    o.m = gml_op(m[0], m[1], m[2], m[3]);
    o.l = gml_op(l[0], l[1], l[2], l[3]);
    return o;
}

// tanh of two arguments that arrive pre-scaled by 2 log2(e) (z = x / k, k = ln(2) / 2).  TA: relative-accurate -- below |x| = 1/4 the odd
// series of gml_tanh_small in the scaled argument, x + x^3 P(x^2) = z (k + w (b0 + w (b1 + w (b2 + w b3)))), w = z^2,
// b_i = k^(2 i + 3) a_i, both values at once on the packed fp32 pipe (6 packed operations per pair; series error 8e-9 relative);
// otherwise (and above 1/4) the short form 1 - 2 / (e^2x + 1), ~2e-7 ABSOLUTE
template <bool TA>
__device__ __forceinline__ void gml_tanh_pair_scaled(float z2, float z3, float& t2, float& t3) {
    t2 = fmaf(-2.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z2) + 1.f), 1.f);
    t3 = fmaf(-2.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z3) + 1.f), 1.f);
    if constexpr (TA) {
        const f32x2 zz = f32x2{z2, z3};
        const f32x2 w = zz * zz;
        f32x2 q = w * 1.5776033314321529e-06f - 3.241205933362716e-05f;     // b3 = k^9 62/2835, b2 = -k^7 17/315
        q = q * w + 6.666779073214221e-04f;                               // b1 = k^5 2/15
        q = q * w - 1.3876027166205392e-02f;                              // b0 = -k^3 / 3
        q = q * w + 0.34657359027997264f;                                 // k
        q = q * zz;
        t2 = fabsf(z2) < 0.7213475f ? q.x : t2;
        t3 = fabsf(z3) < 0.7213475f ? q.y : t3;
    }
}

template <int S>
struct GmlChain6W {           // weight operands of one wave and one layer (36 registers)
    bf16x8 a1A[3], a1B[3];    // layer 1: W1, W2, W3 rows (k = in-channel); piece per lane group: A (h, h, m, m), B (h, l, m, l)
    bf16x8 a2[3];             // layer 2: W4 pieces h, m, l (k = [h1 4g..4g+3 | h23 4g..4g+3]); rows 8..15 repeat rows 0..7
};

// (xa, xb) in D layout -> the three piece tuples (a01, a23, b01, b23); the two residual subtractions run on the matrix pipe
// (x - h = [-I | 0] H + x, gml_split_pair: exact) -- the kernel is VALU-issue-bound, an MFMA costs two VALU slots, the 24 unpack /
// subtract operations it replaces cost 24
__device__ __forceinline__ void gml_split3_tiles(const GmlNegI& N, const f32x4 xa, const f32x4 xb, u32x4& ph, u32x4& pm, u32x4& pl) {
    ph = u32x4{gml_pack2(xa[0], xa[1]), gml_pack2(xa[2], xa[3]), gml_pack2(xb[0], xb[1]), gml_pack2(xb[2], xb[3])};
    const bf16x8 H = __builtin_bit_cast(bf16x8, ph);
    const f32x4 ra = GML_MFMA(N.first, H, xa), rb = GML_MFMA(N.second, H, xb);
    pm = u32x4{gml_pack2(ra[0], ra[1]), gml_pack2(ra[2], ra[3]), gml_pack2(rb[0], rb[1]), gml_pack2(rb[2], rb[3])};
    const bf16x8 M = __builtin_bit_cast(bf16x8, pm);
    const f32x4 sa = GML_MFMA(N.first, M, ra), sb = GML_MFMA(N.second, M, rb);
    pl = u32x4{gml_pack2(sa[0], sa[1]), gml_pack2(sa[2], sa[3]), gml_pack2(sb[0], sb[1]), gml_pack2(sb[2], sb[3])};
}

template <int S>
__device__ __forceinline__ void gml_chain6_load_weights(GmlChain6W<S>& W, const float* __restrict__ w1, const float* __restrict__ w2,
                                                        const float* __restrict__ w3, const float* __restrict__ w4, int c16, int g) {
    constexpr int H2 = 2 * S, H4 = 4 * S;
    const float* w123[3] = {w1, w2, w3};
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float v[8];
        const float sc = b == 0 ? 1.f : 2.8853900817779268f;     // W2, W3 carry the 2 log2(e) of the tanh (gml_chain_load_fwd_weights)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c16 < H2 && j < S) ? w123[b][c16 * S + j] * sc : 0.f;
        const GmlOp3 p = gml_wop3(v);
        W.a1A[b] = g < 2 ? p.h : p.m;
        W.a1B[b] = g == 0 ? p.h : (g == 2 ? p.m : p.l);
    }
    {
        const int q = c16 & 7;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * g + j;
            const bool ok = (q < S) && (c < H2);
            v[j] = ok ? w4[q * H4 + c] : 0.f;
            v[4 + j] = ok ? w4[q * H4 + H2 + c] : 0.f;
        }
        const GmlOp3 p = gml_wop3(v);
        W.a2[0] = p.h; W.a2[1] = p.m; W.a2[2] = p.l;
    }
}

// layer-1 B operands of the lane's edge row: BA = (h, m, h, m), BB = (l, h, l, m) by lane group
__device__ __forceinline__ void gml_chain6_b1(const float (&e)[8], int g, bf16x8& BA, bf16x8& BB) {
    uint32_t a[4], b[4];
    const bool odd = g & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t h, m, l;
        gml_split3_pair(e[2 * j], e[2 * j + 1], h, m, l);
        a[j] = odd ? m : h;
        b[j] = g == 1 ? h : (g == 3 ? m : l);
    }
    BA = gml_op(a[0], a[1], a[2], a[3]);
    BB = gml_op(b[0], b[1], b[2], b[3]);
}

// out (pre-activation, rows q = 4 (g & 1) + r) of one tile.  TA: relative-accurate tanh (series below 1/4, as gml_tanh) instead of the
// short form 1 - 2 / (e^2x + 1), whose ~2e-7 ABSOLUTE error is what is left of the branch's error once the products are exact
template <int S, bool TA, bool RES>
__device__ __forceinline__ f32x4 gml_chain6_forward(const GmlChain6W<S>& W, const GmlNegI& negI, const bf16x8 BA, const bf16x8 BB) {
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 z[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) z[b] = GML_MFMA(W.a1A[b], BA, GML_MFMA(W.a1B[b], BB, zero));     // small products first
    float h[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {                            // z[1], z[2] arrive pre-scaled by 2 log2(e): z = x / k, k = ln(2) / 2
        float t2, t3;
        gml_tanh_pair_scaled<TA>(z[1][r], z[2][r], t2, t3);
        h[r] = gml_relu1(z[0][r]);
        h[4 + r] = t2 * t3;
    }
    bf16x8 Bh, Bm, Bl;                                       // (h1 pair, h1 pair, h23 pair, h23 pair) of each piece
    if constexpr (RES) {
        u32x4 ph, pm, pl;
        gml_split3_tiles(negI, f32x4{h[0], h[1], h[2], h[3]}, f32x4{h[4], h[5], h[6], h[7]}, ph, pm, pl);
        Bh = __builtin_bit_cast(bf16x8, ph); Bm = __builtin_bit_cast(bf16x8, pm); Bl = __builtin_bit_cast(bf16x8, pl);
    } else {
        uint32_t hh[4], hm[4], hl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gml_split3_pair(h[2 * j], h[2 * j + 1], hh[j], hm[j], hl[j]);
        Bh = gml_op(hh[0], hh[1], hh[2], hh[3]); Bm = gml_op(hm[0], hm[1], hm[2], hm[3]); Bl = gml_op(hl[0], hl[1], hl[2], hl[3]);
    }
    f32x4 o = GML_MFMA(W.a2[2], Bh, zero);
    o = GML_MFMA(W.a2[1], Bm, o);
    o = GML_MFMA(W.a2[0], Bl, o);
    o = GML_MFMA(W.a2[1], Bh, o);
    o = GML_MFMA(W.a2[0], Bm, o);
    return GML_MFMA(W.a2[0], Bh, o);
}

#ifndef GML_CHAIN6_RES
#define GML_CHAIN6_RES true
#endif

template <int L>
struct GmlChain6Stack {
    const float* w1[L]; const float* w2[L]; const float* w3[L]; const float* w4[L];
    float* out[L];
};

// L layers' edge branches over the SAME raw supports in one pass (L = 1: the single-layer forward, optionally with the second,
// scattered copy out_t[tpos[e]] = out[e] of the dual-order scheme)
template <int S, int L, bool DUAL, bool TA>
__global__ __launch_bounds__(256, 2) void gml_k_edge_chain6_fwd(const float* __restrict__ ea, const GmlChain6Stack<L> a,
                                                               const int32_t* __restrict__ tpos, float* __restrict__ out_t,
                                                               int64_t E, int64_t ntiles) {
    static_assert(!DUAL || L == 1, "the scattered second copy belongs to the single-layer form");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    GmlChain6W<S> W[L];
#pragma unroll
    for (int l = 0; l < L; ++l) gml_chain6_load_weights<S>(W[l], a.w1[l], a.w2[l], a.w3[l], a.w4[l], c16, g);
    GmlNegI negI;
    gml_chain_make_negI(negI, c16, g);
    const int q0 = 4 * (g & 1);
    const int64_t stride = (int64_t)gridDim.x * 8;
    int64_t t = ((int64_t)blockIdx.x * 4 + wave) * 2;
    const auto rs_t = __builtin_amdgcn_make_buffer_rsrc(DUAL ? out_t : a.out[0], 0, DUAL ? (int)(uint32_t)(E * S * 4) : 0, 0x00020000);
    float en[2][8];
    int32_t tpn[2] = {0, 0};
    auto fetch = [&](int64_t tt) {                             // unconditional, clamped: the loads stay countable
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t e = min((tt + u) * 16 + c16, E - 1);
            const float* p = ea + e * S;
            if constexpr (S % 4 == 0) {
#pragma unroll
                for (int j = 0; j < S / 4; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * j);
                    en[u][4 * j] = v.x; en[u][4 * j + 1] = v.y; en[u][4 * j + 2] = v.z; en[u][4 * j + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < S; ++j) en[u][j] = p[j];
            }
#pragma unroll
            for (int j = S; j < 8; ++j) en[u][j] = 0.f;
            if constexpr (DUAL) tpn[u] = tpos[e];
        }
    };
    float ec[2][8];
    int32_t tp[2] = {0, 0};
    auto take = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < S; ++j) { asm volatile("" : "+v"(en[u][j])); ec[u][j] = en[u][j]; }
#pragma unroll
            for (int j = S; j < 8; ++j) ec[u][j] = 0.f;
            if constexpr (DUAL) { asm volatile("" : "+v"(tpn[u])); tp[u] = tpn[u]; }
        }
    };
    fetch(t);
    take();
    for (; t < ntiles; t += stride) {
        bf16x8 BA[2], BB[2];
        bool valid[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            valid[u] = (t + u) * 16 + c16 < E;
            gml_chain6_b1(ec[u], g, BA[u], BB[u]);             // (lanes past E carry the last edge's row: their stores are dropped)
        }
        fetch(t + stride);                                     // next pair in flight during this pair's chains
        __builtin_amdgcn_sched_barrier(0);
        const int64_t tb = t * (16 * S * 4);                   // wave-uniform byte offset of the pair in every output
        const uint32_t tlo = __builtin_amdgcn_readfirstlane((uint32_t)tb), thi = __builtin_amdgcn_readfirstlane((uint32_t)(tb >> 32));
        const int64_t left = E - t * 16;
        const int nrec = __builtin_amdgcn_readfirstlane((int)(left < 32 ? left : 32)) * (S * 4);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            f32x4 o[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) o[u] = gml_chain6_forward<S, TA, GML_CHAIN6_RES>(W[l], negI, BA[u], BB[u]);
            // lane groups 0,1 hold q = 0..3 / 4..7 and write `out`; groups 2,3 hold the same rows again and (DUAL) write the second,
            // source-sorted copy at tpos[e]; the range check of the descriptor drops edges past E
            const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out[l]) + (((uint64_t)thi << 32) | tlo), 0, nrec, 0x00020000);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int off_o = (g < 2) ? ((u * 16 + c16) * S + q0) * 4 : (int)0xffffff00;
                const int off_t = (DUAL && g >= 2 && valid[u]) ? (tp[u] * S + q0) * 4 : (int)0xffffff00;
                if constexpr (S % 4 == 0) {
                    const f32x4 v = f32x4{gml_relu1(o[u][0]), gml_relu1(o[u][1]), gml_relu1(o[u][2]), gml_relu1(o[u][3])};
                    const bool q_ok = q0 < S;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, q_ok ? off_o : (int)0xffffff00, 0, 2);
                    if constexpr (DUAL) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_t, q_ok ? off_t : (int)0xffffff00, 0, 0);
                } else if constexpr (S == 6 && !DUAL) {         // rows of 24 bytes: columns 0..3 as one 16-byte store, 4, 5 as one 8-byte store
                    typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
                    const f32x4 v = f32x4{gml_relu1(o[u][0]), gml_relu1(o[u][1]), gml_relu1(o[u][2]), gml_relu1(o[u][3])};
                    if (q0 == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, off_o, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b64(u32x2_{__float_as_uint(v[0]), __float_as_uint(v[1])}, rs_o, off_o, 0, 2);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t v = __float_as_uint(gml_relu1(o[u][r]));
                        const bool q_ok = q0 + r < S;
                        __builtin_amdgcn_raw_buffer_store_b32(v, rs_o, q_ok ? off_o + 4 * r : (int)0xffffff00, 0, 0);
                        if constexpr (DUAL) __builtin_amdgcn_raw_buffer_store_b32(v, rs_t, q_ok ? off_t + 4 * r : (int)0xffffff00, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        take();                                                // the wait for the prefetch belongs HERE (exact count)
    }
}

// GML_CHAIN6_TANH=0 in the environment: the short tanh (A/B)
static inline bool gml_chain6_accurate_tanh() {
    static const bool v = [] { const char* e = getenv("GML_CHAIN6_TANH"); return !(e && e[0] == '0'); }();
    return v;
}

template <int S, int L>
int gml_launch_edge_chain6_fwd(const float* ea, const GmlChain6Stack<L>& a, const int32_t* tpos, float* out_t, int64_t E, hipStream_t st) {
    const int64_t ntiles = gml_cdiv(E, 16);
    int64_t grid = gml_cdiv(ntiles, 8);
    if (grid > 4 * GML_NUM_CU) grid = 4 * GML_NUM_CU;
    const dim3 gd((unsigned)grid), bd(256);
    const bool ta = gml_chain6_accurate_tanh();
    if constexpr (L == 1) {
        if (out_t != nullptr) {
            if (ta) hipLaunchKernelGGL((gml_k_edge_chain6_fwd<S, 1, true, true>), gd, bd, 0, st, ea, a, tpos, out_t, E, ntiles);
            else hipLaunchKernelGGL((gml_k_edge_chain6_fwd<S, 1, true, false>), gd, bd, 0, st, ea, a, tpos, out_t, E, ntiles);
            return gml_launch_status();
        }
    }
    if (ta) hipLaunchKernelGGL((gml_k_edge_chain6_fwd<S, L, false, true>), gd, bd, 0, st, ea, a, tpos, out_t, E, ntiles);
    else hipLaunchKernelGGL((gml_k_edge_chain6_fwd<S, L, false, false>), gd, bd, 0, st, ea, a, tpos, out_t, E, ntiles);
    return gml_launch_status();
}
