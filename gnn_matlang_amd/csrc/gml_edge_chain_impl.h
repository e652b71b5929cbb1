// ML3Layer edge branch on the bf16 matrix cores, S = Sout <= 8 (reference: /root/reference/libs/spect_conv.py:190-194,
// 205-207).  Same function as gml_edge_mlp_impl.h, different machine mapping:
//
//   out = relu( W4 . [ relu(W1 e) ; tanh(W2 e) * tanh(W3 e) ] )
//
// A wave works on tiles of 16 edges.  The edge index is ALWAYS the column (lane & 15) of a 16x16 MFMA tile and
// the channel index the row, so the D registers of one v_mfma_f32_16x16x32_bf16 (lane (col, g = lane >> 4)
// holds rows 4g..4g+3) are, after the elementwise step, directly the B operand of the next one: the k-slot
// order of an MFMA is free as long as A (the weights, arranged once per wave in registers) uses the same
// order.  No LDS, no scalar weight stream, nothing but e / gout is read per edge.
//
// fp32-class accuracy from bf16 inputs: every fp32 value is split x = hi + lo (two bf16, residual 2^-17 |x|)
// and the split products are summed in the fp32 accumulator.  Where the contraction is short (K = S <= 8:
// layer 1 and the W4^T back-projection) the four split products hi.hi, hi.lo, lo.hi, lo.lo occupy the four
// 8-slot lane groups of ONE K = 32 instruction.
//
// Weight gradients contract over EDGES, i.e. over the column index: the tiles are transposed on the matrix
// cores (tile^T = [hi | lo] x [I ; I], exact to the split residual), two transposed tiles (32 edges) form the
// K = 32 operands of  dW += [h1; h23; gz1; gz2; gz3]^T-tiles x [go | e]^T-tile,  accumulated in registers over
// the wave's whole edge range; one partial per workgroup, folded by gml_k_reduce_partials in fixed order.
#pragma once
#include "gml_common.h"


__device__ __forceinline__ uint32_t gml_pack2(float a, float b) {           // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float gml_bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float gml_bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// x[0..3] -> packed bf16 pairs hi[2], lo[2] with hi + lo = x to 2^-17 relative
__device__ __forceinline__ void gml_split4(const float (&x)[4], uint32_t (&hi)[2], uint32_t (&lo)[2]) {
    hi[0] = gml_pack2(x[0], x[1]);
    hi[1] = gml_pack2(x[2], x[3]);
    lo[0] = gml_pack2(x[0] - gml_bf_lo(hi[0]), x[1] - gml_bf_hi(hi[0]));
    lo[1] = gml_pack2(x[2] - gml_bf_lo(hi[1]), x[3] - gml_bf_hi(hi[1]));
}
__device__ __forceinline__ void gml_split4v(const f32x4 x, uint32_t (&hi)[2], uint32_t (&lo)[2]) {
    const float t[4] = {x[0], x[1], x[2], x[3]};
    gml_split4(t, hi, lo);
}
__device__ __forceinline__ bf16x8 gml_op(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_bit_cast(bf16x8, u32x4{a, b, c, d});
}
// 8 slot values of a weight operand -> the hi or the lo image
__device__ __forceinline__ bf16x8 gml_wop(const float (&v)[8], bool lo_image) {
    uint32_t h0[2], l0[2], h1[2], l1[2];
    const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
    gml_split4(a, h0, l0);
    gml_split4(b, h1, l1);
    return lo_image ? gml_op(l0[0], l0[1], l1[0], l1[1]) : gml_op(h0[0], h0[1], h1[0], h1[1]);
}
#define GML_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// Splitting is the VALU cost of this design (convert, unpack, subtract, convert per value); the subtraction is moved
// to the matrix pipe: for two tiles xa, xb in D layout, H = (hi(xa) pairs, hi(xb) pairs) is at once the B operand
// k-slots [tile a rows | tile b rows], so  xa - hi(xa) = [-I | 0] x H + xa  and  xb - hi(xb) = [0 | -I] x H + xb  are two
// MFMAs with constant A operands (exact: the products are single bf16 values, the sum is representable).
struct GmlNegI {
    bf16x8 first, second;     // [-I | 0], [0 | -I]: A[row][slot (g, j)] = -1 iff row = 4g + (j & 3) in that half
};
__device__ __forceinline__ void gml_chain_make_negI(GmlNegI& N, int c16, int g) {
    float va[8], vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float d = (c16 == 4 * g + (j & 3)) ? -1.f : 0.f;
        va[j] = j < 4 ? d : 0.f;
        vb[j] = j < 4 ? 0.f : d;
    }
    N.first = gml_wop(va, false);
    N.second = gml_wop(vb, false);
}
// (xa, xb) -> hi = (a01, a23, b01, b23), lo likewise
__device__ __forceinline__ void gml_split_pair(const GmlNegI& N, const f32x4 xa, const f32x4 xb, u32x4& hi, u32x4& lo) {
    hi = u32x4{gml_pack2(xa[0], xa[1]), gml_pack2(xa[2], xa[3]), gml_pack2(xb[0], xb[1]), gml_pack2(xb[2], xb[3])};
    const bf16x8 H = __builtin_bit_cast(bf16x8, hi);
    const f32x4 ra = GML_MFMA(N.first, H, xa);
    const f32x4 rb = GML_MFMA(N.second, H, xb);
    lo = u32x4{gml_pack2(ra[0], ra[1]), gml_pack2(ra[2], ra[3]), gml_pack2(rb[0], rb[1]), gml_pack2(rb[2], rb[3])};
}
// one tile -> [hi | lo] (the k-slot layout of the short-contraction operands and of the transposes)
__device__ __forceinline__ u32x4 gml_split_one(const GmlNegI& N, const f32x4 x) {
    u32x4 t = u32x4{gml_pack2(x[0], x[1]), gml_pack2(x[2], x[3]), 0u, 0u};
    const f32x4 r = GML_MFMA(N.first, __builtin_bit_cast(bf16x8, t), x);
    t.z = gml_pack2(r[0], r[1]);
    t.w = gml_pack2(r[2], r[3]);
    return t;
}

template <int S>
struct GmlChainW {            // weight operands of one wave (registers)
    bf16x8 a1[3];             // layer 1: W1, W2, W3 rows (k = in-channel; groups 0,1 hi / 2,3 lo)
    bf16x8 a2h, a2l;          // layer 2: W4 (k = [h1 4g..4g+3 | h23 4g..4g+3]); rows 8..15 repeat rows 0..7
    GmlNegI negI;
};

template <int S>
__device__ __forceinline__ void gml_chain_load_fwd_weights(GmlChainW<S>& W, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, const float* __restrict__ w3,
                                                           const float* __restrict__ w4, int c16, int g) {
    constexpr int H2 = 2 * S, H4 = 4 * S;
    const float* w123[3] = {w1, w2, w3};
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float v[8];
        // W2, W3 carry the 2 log2(e) of tanh(z) = 1 - 2 / (2^(2 log2(e) z) + 1): one multiply less per tanh
        const float sc = b == 0 ? 1.f : 2.8853900817779268f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c16 < H2 && j < S) ? w123[b][c16 * S + j] * sc : 0.f;
        W.a1[b] = gml_wop(v, g >= 2);
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
        W.a2h = gml_wop(v, false);
        W.a2l = gml_wop(v, true);
    }
    gml_chain_make_negI(W.negI, c16, g);
}

// per-tile forward state kept for the backward
struct GmlChainT {
    float e[8];               // the lane's edge row (all 4 lane groups of a column hold the same row)
    f32x4 z1, t2, t3;         // W1 e ; tanh(W2 e) ; tanh(W3 e)      rows 4g..4g+3
    u32x4 hh, hl;             // split [relu(z1) | t2*t3]: (h1 pair, h1 pair, h23 pair, h23 pair), hi and lo images
    f32x4 out;                // W4 h (pre-activation), row r <-> q = 4(g&1) + r
};

template <int S>
__device__ __forceinline__ void gml_chain_load_e(const float* __restrict__ ea, int64_t eid, bool valid, float (&e)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = 0.f;
    if (valid) {
        const float* p = ea + eid * S;
        if constexpr (S % 4 == 0) {
#pragma unroll
            for (int j = 0; j < S / 4; ++j) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * j);
                e[4 * j] = t.x; e[4 * j + 1] = t.y; e[4 * j + 2] = t.z; e[4 * j + 3] = t.w;
            }
        } else if constexpr (S % 2 == 0) {
#pragma unroll
            for (int j = 0; j < S / 2; ++j) {
                const f32x2 t = *reinterpret_cast<const f32x2*>(p + 2 * j);
                e[2 * j] = t.x; e[2 * j + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < S; ++j) e[j] = p[j];
        }
    }
}

// RES: residuals of the h split on the matrix pipe (backward kernel, VALU-issue-bound) or on the VALU (forward kernel,
// store-bound: the extra MFMA round trip only lengthens its dependency chain).  Both give the same bits (the
// subtraction is exact either way), so the relu mask recomputed by the backward matches the forward's.
// layer 1 operand from the fp32 row: even lane groups carry hi(e), odd ones lo(e)  ->  (Whi + Wlo)(ehi + elo) in one MFMA
__device__ __forceinline__ bf16x8 gml_chain_b1(const float (&e)[8], int g) {
    uint32_t b1[4];
    const bool odd = g & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x0 = e[2 * j], x1 = e[2 * j + 1];
        const float t0 = __uint_as_float(__float_as_uint(x0) & 0xffff0000u);
        const float t1 = __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        b1[j] = gml_pack2(odd ? x0 - t0 : t0, odd ? x1 - t1 : t1);
    }
    return gml_op(b1[0], b1[1], b1[2], b1[3]);
}

// The raw supports are per-batch constants, so their split can be made once (gml_edge_presplit): per edge 32 bytes,
// hi[8] then lo[8] (bf16, the same truncate / round-the-residual split as gml_chain_b1, channels >= S zero).  A lane
// then loads its layer-1 operand with one 16-byte load and no arithmetic.
__global__ __launch_bounds__(256) void gml_k_edge_presplit(const float* __restrict__ ea, uint32_t* __restrict__ es,
                                                          int64_t E, int S);

template <int S, bool RES>
__device__ __forceinline__ void gml_chain_forward(const GmlChainW<S>& W, GmlChainT& T, const bf16x8 B1, int g) {
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    T.z1 = GML_MFMA(W.a1[0], B1, zero);
    const f32x4 z2 = GML_MFMA(W.a1[1], B1, zero);
    const f32x4 z3 = GML_MFMA(W.a1[2], B1, zero);
    f32x4 h1, h23;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                            // z2, z3 arrive pre-scaled by 2 log2(e)
#if defined(GML_EABL) && (GML_EABL & 4)
        T.t2[r] = z2[r]; T.t3[r] = z3[r];
#else
#if defined(GML_CHAIN_TANH) && GML_CHAIN_TANH == 1
        // experiment (round 5): relative-accurate tanh in the chain (series below 1/4, (e - 1) / (e + 1) above): the learned supports'
        // error 4.6e-7 -> 4.0e-7 rms (the bf16 splits dominate it), edge forward 0.89 -> 1.45, backward 2.02 -> 2.52 ms/step: not taken
        {
            const float x2_ = z2[r] * 0.34657359027997264f, x3_ = z3[r] * 0.34657359027997264f;   // back from the 2 log2(e) scale
            const float e2_ = __builtin_amdgcn_exp2f(z2[r]), e3_ = __builtin_amdgcn_exp2f(z3[r]);
            const float b2_ = (e2_ - 1.f) * __builtin_amdgcn_rcpf(e2_ + 1.f), b3_ = (e3_ - 1.f) * __builtin_amdgcn_rcpf(e3_ + 1.f);
            T.t2[r] = fabsf(x2_) < 0.25f ? gml_tanh_small(x2_) : (e2_ > 3.0e38f ? 1.f : b2_);
            T.t3[r] = fabsf(x3_) < 0.25f ? gml_tanh_small(x3_) : (e3_ > 3.0e38f ? 1.f : b3_);
        }
#else
        T.t2[r] = fmaf(-2.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z2[r]) + 1.f), 1.f);
        T.t3[r] = fmaf(-2.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z3[r]) + 1.f), 1.f);
#endif
#endif
        h1[r] = fmaxf(T.z1[r], 0.f);
        h23[r] = T.t2[r] * T.t3[r];
    }
    if constexpr (RES) {
        gml_split_pair(W.negI, h1, h23, T.hh, T.hl);
    } else {
        uint32_t h1h[2], h1l[2], h23h[2], h23l[2];
        gml_split4v(h1, h1h, h1l);
        gml_split4v(h23, h23h, h23l);
        T.hh = u32x4{h1h[0], h1h[1], h23h[0], h23h[1]};
        T.hl = u32x4{h1l[0], h1l[1], h23l[0], h23l[1]};
    }
    const bf16x8 B2h = __builtin_bit_cast(bf16x8, T.hh);
    const bf16x8 B2l = __builtin_bit_cast(bf16x8, T.hl);
    f32x4 o = GML_MFMA(W.a2l, B2h, zero);
    o = GML_MFMA(W.a2h, B2l, o);
    T.out = GML_MFMA(W.a2h, B2h, o);
}

// ------------------------------------------------------------------------------------------ forward kernel
template <int S, bool PRE>
__global__ __launch_bounds__(256, 2) void gml_k_edge_chain_fwd(const float* __restrict__ ea, const uint32_t* __restrict__ es,
                                                              const float* __restrict__ w1,
                                                              const float* __restrict__ w2, const float* __restrict__ w3,
                                                              const float* __restrict__ w4, float* __restrict__ out,
                                                              const int32_t* __restrict__ tpos, float* __restrict__ out_t,
                                                              int64_t E, int64_t ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    GmlChainW<S> W;
    gml_chain_load_fwd_weights<S>(W, w1, w2, w3, w4, c16, g);
    const int q0 = 4 * (g & 1);
    const int64_t stride = (int64_t)gridDim.x * 8;
    int64_t t = ((int64_t)blockIdx.x * 4 + wave) * 2;
    // Stores go through buffer descriptors whose range check drops the lanes that have nothing to write (edges
    // past E, q >= S): no predicate, so the compiler counts them and the wait for the NEXT tile pair's operands
    // (prefetched below, before this pair's arithmetic) does not also wait for these stores to drain.
    // `out`: descriptor based at the wave's tile pair (offsets stay tiny); `out_t`: one descriptor over the array
    // (the dispatcher refuses the second order when E * S * 4 does not fit a 32-bit offset).
    const auto rs_t = __builtin_amdgcn_make_buffer_rsrc(out_t != nullptr ? out_t : out, 0,
                                                        out_t != nullptr ? (int)(uint32_t)(E * S * 4) : 0, 0x00020000);
    const int32_t* tposb = out_t != nullptr ? tpos : reinterpret_cast<const int32_t*>(ea);   // always a readable array of >= E words
    u32x4 b1n[2];
    int32_t tpn[2] = {0, 0};
    auto fetch = [&](int64_t tt) {                             // unconditional, clamped: the loads stay countable
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t e = min((tt + u) * 16 + c16, E - 1);
            b1n[u] = *reinterpret_cast<const u32x4*>(es + e * 8 + 4 * (g & 1));
            tpn[u] = tposb[e];
        }
    };
    u32x4 b1[2];                                               // the current pair's operands: taken over from the prefetch
    int32_t tp[2] = {0, 0};                                    // registers at the END of the previous trip, where the wait
    if constexpr (PRE) {                                       // is an exact count (at the loop top it would merge with
        fetch(t);                                              // the store-less entry path into vmcnt(0))
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            asm volatile("" : "+v"(b1n[u]), "+v"(tpn[u]));
            b1[u] = b1n[u]; tp[u] = tpn[u];
        }
    }
    for (; t < ntiles; t += stride) {
        GmlChainT T[2];
        int64_t eid[2];
        bool valid[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            eid[u] = (t + u) * 16 + c16;
            valid[u] = eid[u] < E;
            if constexpr (PRE) {
                if (!valid[u]) b1[u] = u32x4{0u, 0u, 0u, 0u};
            } else {
                gml_chain_load_e<S>(ea, eid[u], valid[u], T[u].e);
                tp[u] = tposb[min(eid[u], E - 1)];
            }
        }
        if constexpr (PRE) {
            fetch(t + stride);                                 // next pair in flight during this pair's chain
            __builtin_amdgcn_sched_barrier(0);                 // (the scheduler would sink the loads to the loop's end)
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if constexpr (PRE) gml_chain_forward<S, false>(W, T[u], __builtin_bit_cast(bf16x8, b1[u]), g);
            else gml_chain_forward<S, false>(W, T[u], gml_chain_b1(T[u].e, g), g);
        }
        // lane groups 0,1 hold q = 0..3 / 4..7 and write `out`; groups 2,3 hold the same rows again and write the
        // second (source-sorted) copy at tpos[e]
        const int64_t tb = t * (16 * S * 4);                   // wave-uniform byte offset of the pair in `out`
        const uint32_t tlo = __builtin_amdgcn_readfirstlane((uint32_t)tb), thi = __builtin_amdgcn_readfirstlane((uint32_t)(tb >> 32));
        const int64_t left = E - t * 16;                       // edges from the pair's first one (wave-uniform)
        const int nrec = __builtin_amdgcn_readfirstlane((int)(left < 32 ? left : 32)) * (S * 4);
        const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(out) + (((uint64_t)thi << 32) | tlo), 0, nrec, 0x00020000);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int off_o = (g < 2) ? ((u * 16 + c16) * S + q0) * 4 : (int)0xffffff00;
            const int off_t = (g >= 2 && valid[u]) ? (tp[u] * S + q0) * 4 : (int)0xffffff00;
            if constexpr (S % 4 == 0) {
                const f32x4 v = f32x4{fmaxf(T[u].out[0], 0.f), fmaxf(T[u].out[1], 0.f), fmaxf(T[u].out[2], 0.f),
                                      fmaxf(T[u].out[3], 0.f)};
                const bool q_ok = q0 < S;
#ifndef GML_FABL
#define GML_FABL 0
#endif
#if !(GML_FABL & 2)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, q_ok ? off_o : (int)0xffffff00, 0, /*nt: written once, read by the next kernel from HBM anyway*/ 2);
#else
                asm volatile("" :: "v"(v));
#endif
#if !(GML_FABL & 1)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_t, q_ok ? off_t : (int)0xffffff00, 0, 0);
#endif
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t v = __float_as_uint(fmaxf(T[u].out[r], 0.f));
                    const bool q_ok = q0 + r < S;
                    __builtin_amdgcn_raw_buffer_store_b32(v, rs_o, q_ok ? off_o + 4 * r : (int)0xffffff00, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(v, rs_t, q_ok ? off_t + 4 * r : (int)0xffffff00, 0, 0);
                }
            }
        }
        if constexpr (PRE) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                asm volatile("" : "+v"(b1n[u]), "+v"(tpn[u]));    // the wait for the prefetch belongs HERE (exact count)
                b1[u] = b1n[u]; tp[u] = tpn[u];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ forward of a layer stack
// Every ML3Layer of a stack applies ITS edge branch to the SAME raw supports (Zinc12k.py:338-341 passes data.edge_attr2 to
// conv1 .. conv4): the kernel above is HBM-bound reading them (32 B per edge in, 32 B out), so L launches read them L times.
// Here one pass reads a tile pair's pre-split operand once and runs the L chains on it, each with its own weight registers
// (20 VGPRs per layer) and its own output array: (1 + L) x 32 B per edge instead of 2 L x 32 B.
template <int L>
struct GmlChainStack {
    const float* w1[L]; const float* w2[L]; const float* w3[L]; const float* w4[L];
    float* out[L];
};

template <int S, int L>
__global__ __launch_bounds__(256, 2) void gml_k_edge_chain_fwd_stack(const uint32_t* __restrict__ es, const GmlChainStack<L> a,
                                                                    int64_t E, int64_t ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    GmlChainW<S> W[L];
#pragma unroll
    for (int l = 0; l < L; ++l) gml_chain_load_fwd_weights<S>(W[l], a.w1[l], a.w2[l], a.w3[l], a.w4[l], c16, g);
    const int q0 = 4 * (g & 1);
    const int64_t stride = (int64_t)gridDim.x * 8;
    int64_t t = ((int64_t)blockIdx.x * 4 + wave) * 2;
    u32x4 b1n[2], b1[2];
    auto fetch = [&](int64_t tt) {                             // unconditional, clamped: the loads stay countable
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t e = min((tt + u) * 16 + c16, E - 1);
            b1n[u] = *reinterpret_cast<const u32x4*>(es + e * 8 + 4 * (g & 1));
        }
    };
    fetch(t);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        asm volatile("" : "+v"(b1n[u]));
        b1[u] = b1n[u];
    }
    for (; t < ntiles; t += stride) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if ((t + u) * 16 + c16 >= E) b1[u] = u32x4{0u, 0u, 0u, 0u};
        fetch(t + stride);                                     // next pair in flight during this pair's chains
        __builtin_amdgcn_sched_barrier(0);
        const int64_t tb = t * (16 * S * 4);                   // wave-uniform byte offset of the pair in every output
        const uint32_t tlo = __builtin_amdgcn_readfirstlane((uint32_t)tb), thi = __builtin_amdgcn_readfirstlane((uint32_t)(tb >> 32));
        const int64_t left = E - t * 16;
        const int nrec = __builtin_amdgcn_readfirstlane((int)(left < 32 ? left : 32)) * (S * 4);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            GmlChainT T[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) gml_chain_forward<S, false>(W[l], T[u], __builtin_bit_cast(bf16x8, b1[u]), g);
            // lane groups 0,1 hold q = 0..3 / 4..7; the range check of the descriptor drops edges past E
            const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out[l]) + (((uint64_t)thi << 32) | tlo), 0, nrec, 0x00020000);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int off_o = (g < 2 && q0 < S) ? ((u * 16 + c16) * S + q0) * 4 : (int)0xffffff00;
                if constexpr (S % 4 == 0) {
                    const f32x4 v = f32x4{fmaxf(T[u].out[0], 0.f), fmaxf(T[u].out[1], 0.f), fmaxf(T[u].out[2], 0.f), fmaxf(T[u].out[3], 0.f)};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, off_o, 0, 2);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaxf(T[u].out[r], 0.f)), rs_o,
                                                              (g < 2 && q0 + r < S) ? off_o + 4 * r : (int)0xffffff00, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            asm volatile("" : "+v"(b1n[u]));                   // the wait for the prefetch belongs HERE (exact count)
            b1[u] = b1n[u];
        }
    }
}

template <int S, int L>
int gml_launch_edge_chain_fwd_stack(const uint32_t* es, const GmlChainStack<L>& a, int64_t E, hipStream_t st) {
    const int64_t ntiles = gml_cdiv(E, 16);
    int64_t grid = gml_cdiv(ntiles, 8);
    if (grid > 4 * GML_NUM_CU) grid = 4 * GML_NUM_CU;
    hipLaunchKernelGGL((gml_k_edge_chain_fwd_stack<S, L>), dim3((unsigned)grid), dim3(256), 0, st, es, a, E, ntiles);
    return gml_launch_status();
}

// ------------------------------------------------------------------------------------------ backward kernel
template <int S, bool GIN>
struct GmlChainWB {
    bf16x8 a3[2];             // W4^T blocks (h1 part, h23 part): k = q, groups 0,1 hi / 2,3 lo, slots [q | q]
    bf16x8 a5[GIN ? 4 : 1];   // [W1|W2|W3]^T operands for d e: [H1|H2], [L1|L2], [H3|0], [L3|0]
    bf16x8 bIh, bIl;          // [I ; 0] and [0 ; I]: transposes of the hi / the lo image of a split tile
    bf16x8 aE;                // rows 8..15 <- hi + lo of the pre-split supports (e rows of the [go | e] tile, exact)
};

#define GML_CHAIN_NW(S) (6 * (S) * (S) + 4 * (S) * (S))

// 4 consecutive floats of a row of S starting at channel q0 (zero beyond S / for invalid lanes)
template <int S>
__device__ __forceinline__ void gml_chain_load_q(const float* __restrict__ base, int64_t eid, bool valid, int q0, float (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = 0.f;
    if (valid) {
        const float* gp = base + eid * S + q0;
        if constexpr (S % 4 == 0) {
            if (q0 < S) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(gp);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (q0 + r < S) v[r] = gp[r];
        }
    }
}

template <int S, bool GIN, bool PRE>
__global__ __launch_bounds__(256, 3) void gml_k_edge_chain_bwd(
    const float* __restrict__ ea, const uint32_t* __restrict__ es, const float* __restrict__ w1, const float* __restrict__ w2,
    const float* __restrict__ w3, const float* __restrict__ w4, const float* __restrict__ gout,
    float* __restrict__ gin, float* __restrict__ partial, int64_t E, int64_t ntiles) {
    constexpr int H2 = 2 * S, H4 = 4 * S;
    // 4 waves x 12 row-major bf16 tile images [edge][16 channels] (the transposition scratch of the weight-gradient
    // operands); the same bytes hold the workgroup's partial sums at the end
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 12 * 512];
    float (*red)[20][64] = reinterpret_cast<float (*)[20][64]>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int q0 = 4 * (g & 1);
    unsigned char* trw = smem + wave * (12 * 512);
    // chunk (edge e, channels 4cc..4cc+3) lives at cc * 128 + (e ^ 8 (cc >> 1)) * 8: the 16 lanes of a write group (one cc)
    // cover 128 contiguous bytes, the 32 lanes of a read half (8 edges x 4 cc) 4 distinct 64-byte bank segments
    // (tools/lds_sim.py; the plain [edge][channel] rows made every ds_write_b64 a 4-way bank conflict: 66 % of the LDS
    // cycles in gpurun_out/e6_pmcC)
    const int tr_wo = g * 128 + ((c16 ^ ((g >> 1) << 3)) << 3);                                   // lane (edge c16, cc = g)
    const int tr_ro = (c16 & 3) * 128 + (((4 * g + (c16 >> 2)) ^ (((c16 & 3) >> 1) << 3)) << 3);  // gml_tr_frag: lane (channel c16, edges 4g..)
    GmlChainW<S> W;
    gml_chain_load_fwd_weights<S>(W, w1, w2, w3, w4, c16, g);
    GmlChainWB<S, GIN> WB;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + j;
            v[j] = v[4 + j] = (c16 < H2 && q < S) ? w4[q * H4 + blk * H2 + c16] : 0.f;
        }
        WB.a3[blk] = gml_wop(v, g >= 2);
    }
    if constexpr (GIN) {
        // d e[in] = sum_b sum_z Wb[z][in] gz_b[z]: the gz tiles are split in pairs (gz1, gz2), (gz3, y), so the nine
        // significant products take six K = 32 instructions:  [H1|H2].(hi12, lo12), [L1|L2].hi12, [H3|0].(hi3y, lo3y),
        // [L3|0].hi3y   (the zero half ignores the y slots)
        const int in = c16 & 7;
        float wv[3][4];
        const float* w123[3] = {w1, w2, w3};
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int z = 4 * g + j;
                wv[b][j] = (in < S && z < H2) ? w123[b][z * S + in] : 0.f;
            }
        float v12[8], v3[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v12[j] = wv[0][j]; v12[4 + j] = wv[1][j];
            v3[j] = wv[2][j]; v3[4 + j] = 0.f;
        }
        WB.a5[0] = gml_wop(v12, false);
        WB.a5[1] = gml_wop(v12, true);
        WB.a5[2] = gml_wop(v3, false);
        WB.a5[3] = gml_wop(v3, true);
    }
    {
        float vh[8], vl[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = (c16 == 4 * g + (j & 3)) ? 1.f : 0.f;
            vh[j] = j < 4 ? d : 0.f;
            vl[j] = j < 4 ? 0.f : d;
        }
        WB.bIh = gml_wop(vh, false);
        WB.bIl = gml_wop(vl, false);
        // e = hi + lo on the matrix pipe: lane groups 2, 3 pass (hi[4q..], lo[4q..]) of their edge as k-slots; row 8 + c of D
        // picks both slots of channel c.  The fp32 support rows are then never read (r01 profile: 1.5x the 64 B / edge)
        float ve[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ve[j] = (g >= 2 && c16 == 4 * g + (j & 3)) ? 1.f : 0.f;
        WB.aE = gml_wop(ve, false);
    }
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) acc[b] = zero;

    // transposed split tiles: lane (channel = c16, g) gets [hi | lo] of edges 4g..4g+3 -- exact, the transposed values
    // are bf16 numbers, so the repack is a plain conversion.  The operands are the pair tuples as they are:
    // [I ; 0] selects the first tile of a tuple, [0 ; I] the second.
    auto repack = [&](const f32x4 th, const f32x4 tl) -> bf16x8 {
        return gml_op(gml_pack2(th[0], th[1]), gml_pack2(th[2], th[3]), gml_pack2(tl[0], tl[1]), gml_pack2(tl[2], tl[3]));
    };
    auto transpose_pair = [&](const u32x4 hi, const u32x4 lo, bf16x8& xa, bf16x8& xb) {
        const bf16x8 H = __builtin_bit_cast(bf16x8, hi), L = __builtin_bit_cast(bf16x8, lo);
        xa = repack(GML_MFMA(H, WB.bIh, zero), GML_MFMA(L, WB.bIh, zero));
        xb = repack(GML_MFMA(H, WB.bIl, zero), GML_MFMA(L, WB.bIl, zero));
    };

    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t t = (int64_t)blockIdx.x * 4 + wave;
    // prefetched inputs of the next tile: fp32 row (or, with the presplit buffer, the ready layer-1 operand + the 4 raw
    // values of the [go | e] tile) and 4 values of the gout row
    float e_n[PRE ? 1 : 8], g_n[4];
    u32x4 b1_n = u32x4{0u, 0u, 0u, 0u};
    uint2 eh_n = uint2{0u, 0u}, el_n = uint2{0u, 0u};         // PRE: bf16 hi / lo of channels 4 (g & 1) .. + 3 of the edge's supports
    // PRE: unconditional loads with indices clamped into the arrays (the compiler can count them; a lane outside
    // fetches a valid, unused element and is zeroed when the registers are taken over)
    auto load_q_clamped = [&](const float* base, int64_t en, float (&v)[4]) {
        if constexpr (S % 4 == 0) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(base + en * S + (q0 < S ? q0 : 0));
            v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = base[en * S + (q0 + r < S ? q0 + r : S - 1)];
        }
    };
    auto fetch = [&](int64_t tt) {
        const int64_t en = tt * 16 + c16;
        if constexpr (PRE) {
            const int64_t ec = en < E ? en : E - 1;
            b1_n = *reinterpret_cast<const u32x4*>(es + ec * 8 + 4 * (g & 1));
            eh_n = *reinterpret_cast<const uint2*>(es + ec * 8 + 2 * (g & 1));        // same 32-byte row: no extra HBM traffic
            el_n = *reinterpret_cast<const uint2*>(es + ec * 8 + 4 + 2 * (g & 1));
            load_q_clamped(gout, ec, g_n);
        } else {
            const bool ok = tt < ntiles && en < E;
            gml_chain_load_e<S>(ea, en, ok, e_n);
            gml_chain_load_q<S>(gout, en, ok, q0, g_n);
        }
    };
    // the current tile's inputs; with PRE they are taken over from the prefetch registers at the END of the previous
    // trip, where the wait is an exact count (at the loop top it would merge with the store-less entry path into
    // vmcnt(0) and every tile would wait for the previous tile's gradient store to drain)
    float gq[4], ye[4];
    bf16x8 B1, BE;
    auto take = [&](int64_t tt) {
        asm volatile("" : "+v"(b1_n), "+v"(eh_n), "+v"(el_n));
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(g_n[r]));
        const bool ok = tt * 16 + c16 < E;
        // (channels >= S of the pre-split rows are zero.  Lanes past E keep the clamped edge's finite values: their
        //  gout is zeroed below, so go = 0 and with it every gradient term of the lane)
        B1 = __builtin_bit_cast(bf16x8, b1_n);
        BE = __builtin_bit_cast(bf16x8, u32x4{eh_n.x, eh_n.y, el_n.x, el_n.y});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool okq = ok && q0 + r < S;
            gq[r] = okq ? g_n[r] : 0.f;
        }
    };
    fetch(t);
    if constexpr (PRE) take(t);
    for (; t < ntiles; t += stride) {
        GmlChainT T;
        if constexpr (!PRE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) T.e[j] = e_n[j];
            B1 = gml_chain_b1(T.e, g);
#pragma unroll
            for (int r = 0; r < 4; ++r) ye[r] = (g & 1) ? T.e[4 + r] : T.e[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) gq[r] = g_n[r];
        }
        const int64_t eid = t * 16 + c16;
        const bool valid = eid < E;
        fetch(t + stride);                                     // next tile's rows are in flight while this one is computed
        if constexpr (PRE) __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the loads to the loop's end)
        gml_chain_forward<S, true>(W, T, B1, g);
        if constexpr (PRE) {
            const f32x4 ef = GML_MFMA(WB.aE, BE, zero);        // rows 8..15 (lane groups 2, 3): e = hi + lo, exact
#pragma unroll
            for (int r = 0; r < 4; ++r) ye[r] = ef[r];
        }
        f32x4 go;
#pragma unroll
        for (int r = 0; r < 4; ++r) go[r] = (T.out[r] > 0.f) ? gq[r] : 0.f;
        const bf16x8 B3 = __builtin_bit_cast(bf16x8, gml_split_one(W.negI, go));      // [go hi | go lo]
        const f32x4 dh1 = GML_MFMA(WB.a3[0], B3, zero);
        const f32x4 dh23 = GML_MFMA(WB.a3[1], B3, zero);
        f32x4 gz1, gz2, gz3, y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gz1[r] = (T.z1[r] > 0.f) ? dh1[r] : 0.f;
            gz2[r] = dh23[r] * T.t3[r] * fmaf(-T.t2[r], T.t2[r], 1.f);
            gz3[r] = dh23[r] * T.t2[r] * fmaf(-T.t3[r], T.t3[r], 1.f);
            // [go | e] tile: rows 0..7 = go (lane groups 0,1), rows 8..15 = e (lane groups 2,3)
            y[r] = (g < 2) ? go[r] : ye[r];
        }
        u32x4 g12h, g12l, g3yh, g3yl;
        gml_split_pair(W.negI, gz1, gz2, g12h, g12l);
        gml_split_pair(W.negI, gz3, y, g3yh, g3yl);
        if constexpr (GIN) {
            const bf16x8 H12 = __builtin_bit_cast(bf16x8, g12h), L12 = __builtin_bit_cast(bf16x8, g12l);
            const bf16x8 H3y = __builtin_bit_cast(bf16x8, g3yh), L3y = __builtin_bit_cast(bf16x8, g3yl);
            f32x4 de = GML_MFMA(WB.a5[1], H12, zero);
            de = GML_MFMA(WB.a5[3], H3y, de);
            de = GML_MFMA(WB.a5[0], L12, de);
            de = GML_MFMA(WB.a5[2], L3y, de);
            de = GML_MFMA(WB.a5[0], H12, de);
            de = GML_MFMA(WB.a5[2], H3y, de);
            // rows 4g + r = in-channel (rows 8..15 repeat 0..7).  Store through a buffer descriptor based at the tile:
            // its range check drops the edges past E, an out-of-range offset the lanes with nothing to write --
            // no predicate, so the store is countable (see `take`)
            const int64_t tb = t * (16 * S * 4);
            const uint32_t tlo = __builtin_amdgcn_readfirstlane((uint32_t)tb), thi = __builtin_amdgcn_readfirstlane((uint32_t)(tb >> 32));
            const int64_t left = E - t * 16;
            const int nrec = __builtin_amdgcn_readfirstlane((int)(left < 16 ? left : 16)) * (S * 4);
            const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char*>(gin) + (((uint64_t)thi << 32) | tlo), 0, nrec, 0x00020000);
            const int off = (g < 2) ? (c16 * S + 4 * g) * 4 : (int)0xffffff00;
            if constexpr (S % 4 == 0) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, de), rs_g, (4 * g < S) ? off : (int)0xffffff00, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(de[r]), rs_g,
                                                          (4 * g + r < S) ? off + 4 * r : (int)0xffffff00, 0, 0);
            }
        }
        // weight gradients of the 16 edges: k-slots (g, j < 4) = hi, (g, j >= 4) = lo of edge 4g + (j & 3);
        //   X.[Yh | Yh] = Xh Yh + Xl Yh ,  X.[Yl | 0] = Xh Yl
#ifndef GML_EABL
#define GML_EABL 0
#endif
#if GML_EABL & 2
        asm volatile("" :: "v"(T.hh), "v"(T.hl), "v"(g12h), "v"(g12l), "v"(g3yh), "v"(g3yl));
        if (false) {
#else
        {
#endif
        bf16x8 XT[5], YTb;
#ifdef GML_EDGE_MFMA_TRANSPOSE
        transpose_pair(T.hh, T.hl, XT[0], XT[1]);            // h1, h23
        transpose_pair(g12h, g12l, XT[2], XT[3]);            // gz1, gz2
        transpose_pair(g3yh, g3yl, XT[4], YTb);              // gz3, [go | e]
#else
        // The split tiles are bf16 pairs already (lane (edge, g): channels 4g..4g+3 = 8 bytes per image): written as
        // [edge][channel] images (swizzled, see tr_wo) to the wave's LDS scratch and read back with ds_read_b64_tr_b16, lane (channel, g) receives
        // edges 4g..4g+3 -- the same operands as the matrix-core transposes, without their 12 MFMAs and 24 conversions
        // (an MFMA holds the VALU issue port for 8 cycles, tools/probes/probe_overlap.hip: this kernel is issue-bound).
        // One wave, in-order LDS queue: no barrier.
        {
            const u32x4 im[6] = {T.hh, T.hl, g12h, g12l, g3yh, g3yl};
#pragma unroll
            for (int i = 0; i < 3; ++i) {                    // pair tuple i: tiles 2i (x, y) and 2i + 1 (z, w); hi then lo
                *reinterpret_cast<uint2*>(trw + (4 * i + 0) * 512 + tr_wo) = uint2{im[2 * i].x, im[2 * i].y};
                *reinterpret_cast<uint2*>(trw + (4 * i + 1) * 512 + tr_wo) = uint2{im[2 * i + 1].x, im[2 * i + 1].y};
                *reinterpret_cast<uint2*>(trw + (4 * i + 2) * 512 + tr_wo) = uint2{im[2 * i].z, im[2 * i].w};
                *reinterpret_cast<uint2*>(trw + (4 * i + 3) * 512 + tr_wo) = uint2{im[2 * i + 1].z, im[2 * i + 1].w};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int b = 0; b < 5; ++b) XT[b] = gml_tr_frag(trw + (2 * b) * 512 + tr_ro, trw + (2 * b + 1) * 512 + tr_ro);
            YTb = gml_tr_frag(trw + 10 * 512 + tr_ro, trw + 11 * 512 + tr_ro);
            __builtin_amdgcn_wave_barrier();
        }
#endif
        const u32x4 YT = __builtin_bit_cast(u32x4, YTb);
        const bf16x8 Bhh = gml_op(YT.x, YT.y, YT.x, YT.y);
        const bf16x8 Bl0 = gml_op(YT.z, YT.w, 0u, 0u);
#if GML_EABL & 1
#pragma unroll
        for (int b = 0; b < 5; ++b) asm volatile("" :: "v"(XT[b]), "v"(Bl0), "v"(Bhh));
#else
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            acc[b] = GML_MFMA(XT[b], Bl0, acc[b]);
            acc[b] = GML_MFMA(XT[b], Bhh, acc[b]);
        }
#endif
        }
        if constexpr (PRE) {
            __builtin_amdgcn_sched_barrier(0);
            take(t + stride);
        }
    }

    // one partial per workgroup: fixed-order sum of the 4 waves, then [dw1 (2S*S) | dw2 | dw3 | dw4 (S*4S)]
    __syncthreads();                                          // (the scratch of slower waves is still in use)
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][4 * b + r][lane] = acc[b][r];
    __syncthreads();
    float* P = partial + (int64_t)blockIdx.x * GML_CHAIN_NW(S);
    for (int it = threadIdx.x; it < 20 * 64; it += 256) {
        const int br = it >> 6, ln = it & 63;
        const float v = ((red[0][br][ln] + red[1][br][ln]) + red[2][br][ln]) + red[3][br][ln];
        const int b = br >> 2, row = 4 * (ln >> 4) + (br & 3), col = ln & 15;     // D: row = channel, col = [go | e]
        if (row >= H2) continue;
        if (b < 2) {
            if (col < S) P[6 * S * S + col * H4 + b * H2 + row] = v;               // dW4[q][c]
        } else {
            if (col >= 8 && col < 8 + S) P[(b - 2) * H2 * S + row * S + (col - 8)] = v;   // dWb[z][in]
        }
    }
}

template <int S>
int gml_launch_edge_chain_fwd(const float* ea, const uint32_t* es, const float* w1, const float* w2, const float* w3,
                              const float* w4, float* out, const int32_t* tpos, float* out_t, int64_t E, hipStream_t st);
template <int S>
int gml_launch_edge_chain_bwd(const float* ea, const uint32_t* es, const float* w1, const float* w2, const float* w3, const float* w4,
                              const float* gout, float* gin, float* dw1, float* dw2, float* dw3, float* dw4,
                              int64_t E, void* ws, size_t ws_bytes, hipStream_t st);

__global__ void gml_k_reduce_partials(const float* __restrict__ partial, int64_t nwaves, int nw,
                                      float* __restrict__ d0, int n0, float* __restrict__ d1, int n1,
                                      float* __restrict__ d2, int n2, float* __restrict__ d3, int n3);

// persistent workgroups per CU (GML_EDGE_BWD_WGS, 1..6: 24.5 KB of LDS each); the workspace is sized for the maximum
static inline int gml_edge_chain_bwd_wgs() {
    static const int v = [] { const char* e = getenv("GML_EDGE_BWD_WGS"); const int n = e ? atoi(e) : 6; return n < 1 ? 1 : (n > 6 ? 6 : n); }();
    return v;
}
static inline int64_t gml_edge_chain_bwd_groups(int64_t E, int wgs_per_cu = 8) {
    const int64_t ntiles = gml_cdiv(E, 16);
    int64_t grid = gml_cdiv(ntiles, 4);
    if (grid > wgs_per_cu * GML_NUM_CU) grid = wgs_per_cu * GML_NUM_CU;
    return grid < 1 ? 1 : grid;
}

#define GML_DEFINE_EDGE_CHAIN(SV)                                                                               \
    template <>                                                                                                 \
    int gml_launch_edge_chain_fwd<SV>(const float* ea, const uint32_t* es, const float* w1, const float* w2,    \
                                      const float* w3, const float* w4, float* out, const int32_t* tpos,        \
                                      float* out_t, int64_t E, hipStream_t st) {                                \
        const int64_t ntiles = gml_cdiv(E, 16);                                                                 \
        int64_t grid = gml_cdiv(ntiles, 8);                                                                     \
        static const int fw = [] { const char* e = getenv("GML_EDGE_FWD_WGS"); const int n = e ? atoi(e) : 6; return n < 1 ? 1 : n; }();  \
        if (grid > fw * GML_NUM_CU) grid = fw * GML_NUM_CU;   /* all resident at 70 VGPRs; 4..8 measured within 2 % */   \
        if (es != nullptr)                                                                                      \
            hipLaunchKernelGGL((gml_k_edge_chain_fwd<SV, true>), dim3((unsigned)grid), dim3(256), 0, st, ea,    \
                               es, w1, w2, w3, w4, out, tpos, out_t, E, ntiles);                                \
        else                                                                                                    \
            hipLaunchKernelGGL((gml_k_edge_chain_fwd<SV, false>), dim3((unsigned)grid), dim3(256), 0, st, ea,   \
                               es, w1, w2, w3, w4, out, tpos, out_t, E, ntiles);                                \
        return gml_launch_status();                                                                             \
    }                                                                                                           \
    template <>                                                                                                 \
    int gml_launch_edge_chain_bwd<SV>(const float* ea, const uint32_t* es, const float* w1, const float* w2,    \
                                      const float* w3, const float* w4, const float* gout, float* gin,          \
                                      float* dw1, float* dw2, float* dw3, float* dw4, int64_t E, void* ws,      \
                                      size_t ws_bytes, hipStream_t st) {                                        \
        const int64_t ntiles = gml_cdiv(E, 16);                                                                 \
        const int64_t grid = gml_edge_chain_bwd_groups(E, gml_edge_chain_bwd_wgs());                                                  \
        constexpr int NW = GML_CHAIN_NW(SV);                                                                    \
        if (ws_bytes < (size_t)grid * NW * sizeof(float)) return GML_E_WORKSPACE;                               \
        const dim3 gd((unsigned)grid), bd(256);                                                                 \
        float* wsf = (float*)ws;                                                                                \
        if (gin != nullptr && es != nullptr)                                                                    \
            hipLaunchKernelGGL((gml_k_edge_chain_bwd<SV, true, true>), gd, bd, 0, st, ea, es, w1, w2, w3, w4,   \
                               gout, gin, wsf, E, ntiles);                                                      \
        else if (gin != nullptr)                                                                                \
            hipLaunchKernelGGL((gml_k_edge_chain_bwd<SV, true, false>), gd, bd, 0, st, ea, es, w1, w2, w3, w4,  \
                               gout, gin, wsf, E, ntiles);                                                      \
        else if (es != nullptr)                                                                                 \
            hipLaunchKernelGGL((gml_k_edge_chain_bwd<SV, false, true>), gd, bd, 0, st, ea, es, w1, w2, w3, w4,  \
                               gout, gin, wsf, E, ntiles);                                                      \
        else                                                                                                    \
            hipLaunchKernelGGL((gml_k_edge_chain_bwd<SV, false, false>), gd, bd, 0, st, ea, es, w1, w2, w3, w4, \
                               gout, gin, wsf, E, ntiles);                                                      \
        int rc = gml_launch_status();                                                                           \
        if (rc != GML_OK) return rc;                                                                            \
        if (!dw1) return GML_OK;   /* partials stay in ws: gml_fold_many */                                     \
        const int n123 = 2 * SV * SV, n4 = SV * 4 * SV;                                                         \
        hipLaunchKernelGGL(gml_k_reduce_partials, dim3((unsigned)gml_cdiv(NW, 16)), dim3(256), 0, st,           \
                           (const float*)ws, grid, NW, dw1, n123, dw2, n123, dw3, n123, dw4, n4);               \
        return gml_launch_status();                                                                             \
    }
