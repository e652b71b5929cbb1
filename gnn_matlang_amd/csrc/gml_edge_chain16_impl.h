// ML3Layer edge branch on the bf16 matrix cores for 8 < S = Sout <= 16 (counting.py: S = 12; reference:
// /root/reference/libs/spect_conv.py:190-194, 205-207).  The same function and the same machine mapping as
// gml_edge_chain_impl.h (tiles of 16 edges, edge = column, channel = row, D registers of one MFMA are the B operand of the
// next) with the wider shapes:
//
//   layer 1   W_b [2S x S], 2S <= 32 rows = TWO row tiles; the 16 in-channels take 16 of the 32 k-slots, so the pre-split
//             row of an edge (hi[16] | lo[16], 64 bytes, gml_edge_presplit) IS the B operand (lane group g loads bytes
//             16 g ..) and the split products take two instructions:  [Whi | Whi] . [e_hi | e_lo]  +  [Wlo | 0] . [e_hi | e_lo]
//   layer 2   W4 [S x 4S], K = 4S <= 64 = two K = 32 steps (the h1 tile pair, the h23 tile pair) x three split products
//   backward  d h = W4^T go: four row tiles x ([Whi | Whi] . [go_hi | go_lo] + [Wlo | 0] . [go_hi | 0]);  weight gradients
//             contract over edges: the split tiles (bf16 pairs already) are written to a per-wave LDS scratch and read back
//             transposed with ds_read_b64_tr_b16 (two batches of 12 images through the same 6 KB), 20 MFMAs accumulate
//             [h1; h23]^T x go (dW4) and [gz1; gz2; gz3]^T x e (dW1..3) in registers over the wave's whole edge range.
// The supports' gradient (gin) is not produced here (no reference script trains the raw supports; the dispatcher keeps the
// VALU kernels for that case).
#pragma once
#include "gml_edge_chain_impl.h"

template <int S>
struct GmlChain16W {
    bf16x8 a1a[3][2], a1b[3][2];   // layer 1: [Whi | Whi] and [Wlo | 0] of W1..W3, row tiles 0 / 1
    bf16x8 a2h[2], a2l[2];         // layer 2: W4 hi / lo for the k-steps (h1 tiles) / (h23 tiles)
    GmlNegI negI;
};

template <int S>
__device__ __forceinline__ void gml_chain16_load_fwd_weights(GmlChain16W<S>& W, const float* __restrict__ w1,
                                                             const float* __restrict__ w2, const float* __restrict__ w3,
                                                             const float* __restrict__ w4, int c16, int g) {
    constexpr int H2 = 2 * S, H4 = 4 * S;
    const float* w123[3] = {w1, w2, w3};
    const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float sc = b == 0 ? 1.f : 2.8853900817779268f;      // tanh(z) = 1 - 2 / (2^(2 log2(e) z) + 1)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 16 * t + c16;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = 8 * (g & 1) + j;
                v[j] = (row < H2 && col < S) ? w123[b][row * S + col] * sc : 0.f;
            }
            W.a1a[b][t] = gml_wop(v, false);
            W.a1b[b][t] = g < 2 ? gml_wop(v, true) : gml_wop(zero8, false);
        }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int q = c16;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ca = 4 * g + j, cb = 16 + 4 * g + j;
            v[j] = (q < S && ca < H2) ? w4[q * H4 + st * H2 + ca] : 0.f;
            v[4 + j] = (q < S && cb < H2) ? w4[q * H4 + st * H2 + cb] : 0.f;
        }
        W.a2h[st] = gml_wop(v, false);
        W.a2l[st] = gml_wop(v, true);
    }
    gml_chain_make_negI(W.negI, c16, g);
}

struct GmlChain16T {
    f32x4 z1[2], t2[2], t3[2];     // W1 e ; tanh(W2 e) ; tanh(W3 e): rows 16 t + 4g .. + 3
    u32x4 h1h, h1l, h23h, h23l;    // split tile pairs (tile 0 pair, tile 0 pair, tile 1 pair, tile 1 pair)
    f32x4 out;                     // W4 h (pre-activation), row q = 4g + r
};

template <int S, bool RES>
__device__ __forceinline__ void gml_chain16_forward(const GmlChain16W<S>& W, GmlChain16T& T, const bf16x8 B1) {
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 h1[2], h23[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        T.z1[t] = GML_MFMA(W.a1a[0][t], B1, GML_MFMA(W.a1b[0][t], B1, zero));
        const f32x4 z2 = GML_MFMA(W.a1a[1][t], B1, GML_MFMA(W.a1b[1][t], B1, zero));
        const f32x4 z3 = GML_MFMA(W.a1a[2][t], B1, GML_MFMA(W.a1b[2][t], B1, zero));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            T.t2[t][r] = fmaf(-2.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z2[r]) + 1.f), 1.f);
            T.t3[t][r] = fmaf(-2.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z3[r]) + 1.f), 1.f);
            h1[t][r] = fmaxf(T.z1[t][r], 0.f);
            h23[t][r] = T.t2[t][r] * T.t3[t][r];
        }
    }
    if constexpr (RES) {
        gml_split_pair(W.negI, h1[0], h1[1], T.h1h, T.h1l);
        gml_split_pair(W.negI, h23[0], h23[1], T.h23h, T.h23l);
    } else {
        uint32_t ah[2], al[2], bh[2], bl[2];
        gml_split4v(h1[0], ah, al);
        gml_split4v(h1[1], bh, bl);
        T.h1h = u32x4{ah[0], ah[1], bh[0], bh[1]};
        T.h1l = u32x4{al[0], al[1], bl[0], bl[1]};
        gml_split4v(h23[0], ah, al);
        gml_split4v(h23[1], bh, bl);
        T.h23h = u32x4{ah[0], ah[1], bh[0], bh[1]};
        T.h23l = u32x4{al[0], al[1], bl[0], bl[1]};
    }
    const bf16x8 Ah = __builtin_bit_cast(bf16x8, T.h1h), Al = __builtin_bit_cast(bf16x8, T.h1l);
    const bf16x8 Bh = __builtin_bit_cast(bf16x8, T.h23h), Bl = __builtin_bit_cast(bf16x8, T.h23l);
    f32x4 o = GML_MFMA(W.a2l[0], Ah, zero);
    f32x4 o2 = GML_MFMA(W.a2l[1], Bh, zero);
    o = GML_MFMA(W.a2h[0], Al, o);
    o2 = GML_MFMA(W.a2h[1], Bl, o2);
    o = GML_MFMA(W.a2h[0], Ah, o);
    o2 = GML_MFMA(W.a2h[1], Bh, o2);
    T.out = o + o2;
}

// ------------------------------------------------------------------------------------------ forward kernel
// es: 16 uint32 per edge (hi[16] | lo[16] bf16); out [E, S]; out_t (optional): the same rows at tpos[e]
template <int S>
__global__ __launch_bounds__(256, 2) void gml_k_edge_chain16_fwd(const uint32_t* __restrict__ es, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, const float* __restrict__ w3,
                                                                const float* __restrict__ w4, float* __restrict__ out,
                                                                const int32_t* __restrict__ tpos, float* __restrict__ out_t,
                                                                int64_t E, int64_t ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    GmlChain16W<S> W;
    gml_chain16_load_fwd_weights<S>(W, w1, w2, w3, w4, c16, g);
    const int q0 = 4 * g;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t t = (int64_t)blockIdx.x * 4 + wave;
    auto fetch = [&](int64_t tt, u32x4& b1, int32_t& tp) {      // clamped: always a readable edge
        const int64_t e = min(tt * 16 + c16, E - 1);
        b1 = *reinterpret_cast<const u32x4*>(es + e * 16 + 4 * g);
        tp = out_t != nullptr ? tpos[e] : 0;
    };
    u32x4 b1n;
    int32_t tpn;
    fetch(t, b1n, tpn);
    for (; t < ntiles; t += stride) {
        const u32x4 b1 = b1n;
        const int32_t tp = tpn;
        fetch(t + stride < ntiles ? t + stride : t, b1n, tpn);  // next tile's operand in flight during this chain
        GmlChain16T T;
        gml_chain16_forward<S, false>(W, T, __builtin_bit_cast(bf16x8, b1));
        const int64_t eid = t * 16 + c16;
        if (eid < E && q0 < S) {
            const f32x4 v = f32x4{fmaxf(T.out[0], 0.f), fmaxf(T.out[1], 0.f), fmaxf(T.out[2], 0.f), fmaxf(T.out[3], 0.f)};
            float* o = out + eid * S + q0;
            float* ot = out_t != nullptr ? out_t + (int64_t)tp * S + q0 : nullptr;
            if constexpr (S % 4 == 0) {
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));
                if (ot != nullptr) *reinterpret_cast<f32x4*>(ot) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (q0 + r < S) {
                        o[r] = v[r];
                        if (ot != nullptr) ot[r] = v[r];
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward kernel
template <int S>
struct GmlChain16WB {
    bf16x8 a3a[4], a3b[4];         // W4^T row tiles (h1 0/1, h23 0/1): [Whi | Whi] and [Wlo | 0] over k = q
};

#define GML_CHAIN16_NW(S) (10 * (S) * (S))

// SYM (round 6, gml_edge_chain_sym_impl.h): tile entry u evaluates edge uid[u] on gout[uid[u]] + gout[mir[u]] (mir < 0: alone); E is
// then the number of entries
template <int S, bool SYM = false>
__global__ __launch_bounds__(256, 2) void gml_k_edge_chain16_bwd(const uint32_t* __restrict__ es, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, const float* __restrict__ w3,
                                                                const float* __restrict__ w4, const float* __restrict__ gout,
                                                                float* __restrict__ partial, int64_t E, int64_t ntiles,
                                                                const int32_t* __restrict__ uid = nullptr,
                                                                const int32_t* __restrict__ mir = nullptr) {
    constexpr int H2 = 2 * S, H4 = 4 * S;
    // 4 waves x 12 bf16 tile images (the transposition scratch, see gml_edge_chain_impl.h); the same bytes hold the
    // workgroup's partial sums at the end (5 accumulator tiles at a time)
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 12 * 512];
    float (*red)[20][64] = reinterpret_cast<float (*)[20][64]>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    unsigned char* trw = smem + wave * (12 * 512);
    const int tr_wo = g * 128 + ((c16 ^ ((g >> 1) << 3)) << 3);
    const int tr_ro = (c16 & 3) * 128 + (((4 * g + (c16 >> 2)) ^ (((c16 & 3) >> 1) << 3)) << 3);
    GmlChain16W<S> W;
    gml_chain16_load_fwd_weights<S>(W, w1, w2, w3, w4, c16, g);
    GmlChain16WB<S> WB;
    {
        const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            const int ch = 16 * (tile & 1) + c16, part = tile >> 1;
            float v[8], v2[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 4 * g + j;
                const float x = (ch < H2 && q < S) ? w4[q * H4 + part * H2 + ch] : 0.f;
                v[j] = v[4 + j] = x;
                v2[j] = x;
                v2[4 + j] = 0.f;
            }
            WB.a3a[tile] = gml_wop(v, false);
            WB.a3b[tile] = gml_wop(v2, true);
            (void)zero8;
        }
    }
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[10];
#pragma unroll
    for (int b = 0; b < 10; ++b) acc[b] = zero;

    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t t = (int64_t)blockIdx.x * 4 + wave;
    const int qc = 4 * g < S ? 4 * g : 0;                       // (lanes with no gout column read a valid one, zeroed below)
    u32x4 b1_n;
    uint2 eh_n, el_n;
    f32x4 g_n, g2_n = f32x4{0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int64_t tt) {                              // unconditional, clamped
        int64_t e = min(tt * 16 + c16, E - 1);
        if constexpr (SYM) {
            const int32_t m = mir[e];
            e = uid[e];
            const int64_t mm = m >= 0 ? m : e;                  // (no mirror: a readable row, dropped below)
            if constexpr (S % 4 == 0) g2_n = *reinterpret_cast<const f32x4*>(gout + mm * S + qc);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) g2_n[r] = gout[mm * S + (qc + r < S ? qc + r : S - 1)];
            }
            if (m < 0) g2_n = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const uint32_t* row = es + e * 16;
        b1_n = *reinterpret_cast<const u32x4*>(row + 4 * g);
        eh_n = *reinterpret_cast<const uint2*>(row + 2 * g);        // hi[4g .. 4g+3]
        el_n = *reinterpret_cast<const uint2*>(row + 8 + 2 * g);    // lo[4g .. 4g+3]
        if constexpr (S % 4 == 0) g_n = *reinterpret_cast<const f32x4*>(gout + e * S + qc);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) g_n[r] = gout[e * S + (qc + r < S ? qc + r : S - 1)];
        }
    };
    fetch(t);
    for (; t < ntiles; t += stride) {
        const bf16x8 B1 = __builtin_bit_cast(bf16x8, b1_n);
        const uint2 eh = eh_n, el = el_n;
        const bool ok = t * 16 + c16 < E;
        f32x4 gq;
#pragma unroll
        for (int r = 0; r < 4; ++r) gq[r] = (ok && 4 * g + r < S) ? (SYM ? g_n[r] + g2_n[r] : g_n[r]) : 0.f;
        fetch(t + stride < ntiles ? t + stride : t);
        GmlChain16T T;
        gml_chain16_forward<S, true>(W, T, B1);
        f32x4 go;
#pragma unroll
        for (int r = 0; r < 4; ++r) go[r] = (T.out[r] > 0.f) ? gq[r] : 0.f;
        const u32x4 go_s = gml_split_one(W.negI, go);               // [go hi | go lo]
        const bf16x8 B3 = __builtin_bit_cast(bf16x8, go_s);
        f32x4 dh[4];
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) dh[tile] = GML_MFMA(WB.a3a[tile], B3, GML_MFMA(WB.a3b[tile], B3, zero));
        f32x4 gz1[2], gz2[2], gz3[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gz1[tt][r] = (T.z1[tt][r] > 0.f) ? dh[tt][r] : 0.f;
                gz2[tt][r] = dh[2 + tt][r] * T.t3[tt][r] * fmaf(-T.t2[tt][r], T.t2[tt][r], 1.f);
                gz3[tt][r] = dh[2 + tt][r] * T.t2[tt][r] * fmaf(-T.t3[tt][r], T.t3[tt][r], 1.f);
            }
        u32x4 z1h, z1l, z2h, z2l, z3h, z3l;
        gml_split_pair(W.negI, gz1[0], gz1[1], z1h, z1l);
        gml_split_pair(W.negI, gz2[0], gz2[1], z2h, z2l);
        gml_split_pair(W.negI, gz3[0], gz3[1], z3h, z3l);

        // ---- weight gradients: k-slots (g, j < 4) = hi, (g, j >= 4) = lo of edge 4g + (j & 3);
        //      X.[Yh | Yh] = Xh Yh + Xl Yh ,  X.[Yl | 0] = Xh Yl
        auto put = [&](int slot, uint32_t a, uint32_t b) { *reinterpret_cast<uint2*>(trw + slot * 512 + tr_wo) = uint2{a, b}; };
        auto get = [&](int tile) { return gml_tr_frag(trw + (2 * tile) * 512 + tr_ro, trw + (2 * tile + 1) * 512 + tr_ro); };
        // batch 1: h1 tiles 0/1, h23 tiles 0/1, go, e
        put(0, T.h1h.x, T.h1h.y);   put(1, T.h1l.x, T.h1l.y);   put(2, T.h1h.z, T.h1h.w);   put(3, T.h1l.z, T.h1l.w);
        put(4, T.h23h.x, T.h23h.y); put(5, T.h23l.x, T.h23l.y); put(6, T.h23h.z, T.h23h.w); put(7, T.h23l.z, T.h23l.w);
        put(8, go_s.x, go_s.y);     put(9, go_s.z, go_s.w);     put(10, eh.x, eh.y);        put(11, el.x, el.y);
        __builtin_amdgcn_wave_barrier();
        bf16x8 XT[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) XT[i] = get(i);
        const u32x4 YG = __builtin_bit_cast(u32x4, get(4)), YE = __builtin_bit_cast(u32x4, get(5));
        __builtin_amdgcn_wave_barrier();
        {
            const bf16x8 Bhh = gml_op(YG.x, YG.y, YG.x, YG.y), Bl0 = gml_op(YG.z, YG.w, 0u, 0u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = GML_MFMA(XT[i], Bl0, acc[i]);
                acc[i] = GML_MFMA(XT[i], Bhh, acc[i]);
            }
        }
        // batch 2: gz1 tiles 0/1, gz2 tiles 0/1, gz3 tiles 0/1 (in-order LDS queue: the reads above are done)
        put(0, z1h.x, z1h.y); put(1, z1l.x, z1l.y); put(2, z1h.z, z1h.w);  put(3, z1l.z, z1l.w);
        put(4, z2h.x, z2h.y); put(5, z2l.x, z2l.y); put(6, z2h.z, z2h.w);  put(7, z2l.z, z2l.w);
        put(8, z3h.x, z3h.y); put(9, z3l.x, z3l.y); put(10, z3h.z, z3h.w); put(11, z3l.z, z3l.w);
        __builtin_amdgcn_wave_barrier();
        {
            const bf16x8 Bhh = gml_op(YE.x, YE.y, YE.x, YE.y), Bl0 = gml_op(YE.z, YE.w, 0u, 0u);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bf16x8 X = get(i);
                acc[4 + i] = GML_MFMA(X, Bl0, acc[4 + i]);
                acc[4 + i] = GML_MFMA(X, Bhh, acc[4 + i]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // one partial per workgroup: fixed-order sum of the 4 waves, then [dw1 (2S*S) | dw2 | dw3 | dw4 (S*4S)]
    float* P = partial + (int64_t)blockIdx.x * GML_CHAIN16_NW(S);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();                                        // (scratch of slower waves / the previous pass)
#pragma unroll
        for (int b = 0; b < 5; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][4 * b + r][lane] = acc[5 * pass + b][r];
        __syncthreads();
        for (int it = threadIdx.x; it < 20 * 64; it += 256) {
            const int br = it >> 6, ln = it & 63;
            const float v = ((red[0][br][ln] + red[1][br][ln]) + red[2][br][ln]) + red[3][br][ln];
            const int a = 5 * pass + (br >> 2);                 // accumulator tile
            const int row = 4 * (ln >> 4) + (br & 3), col = ln & 15;
            if (a < 4) {                                        // dW4[q = col][part * 2S + ch], X tile a = (part, t)
                const int ch = 16 * (a & 1) + row, part = a >> 1;
                if (ch < H2 && col < S) P[6 * S * S + col * H4 + part * H2 + ch] = v;
            } else {                                            // dWb[z][in = col], X tile a - 4 = (b, t)
                const int b = (a - 4) >> 1, z = 16 * ((a - 4) & 1) + row;
                if (z < H2 && col < S) P[b * H2 * S + z * S + col] = v;
            }
        }
    }
}

template <int S>
int gml_launch_edge_chain16_fwd(const uint32_t* es, const float* w1, const float* w2, const float* w3, const float* w4,
                                float* out, const int32_t* tpos, float* out_t, int64_t E, hipStream_t st);
template <int S>
int gml_launch_edge_chain16_bwd(const uint32_t* es, const float* w1, const float* w2, const float* w3, const float* w4,
                                const float* gout, float* dw1, float* dw2, float* dw3, float* dw4, int64_t E, void* ws,
                                size_t ws_bytes, hipStream_t st);

// persistent workgroups: 2 per CU (<= 256 VGPRs)
static inline int64_t gml_edge_chain16_bwd_groups(int64_t E) {
    const int64_t ntiles = gml_cdiv(E, 16);
    int64_t grid = gml_cdiv(ntiles, 4);
    if (grid > 2 * GML_NUM_CU) grid = 2 * GML_NUM_CU;
    return grid < 1 ? 1 : grid;
}

#define GML_DEFINE_EDGE_CHAIN16(SV)                                                                              \
    template <>                                                                                                  \
    int gml_launch_edge_chain16_fwd<SV>(const uint32_t* es, const float* w1, const float* w2, const float* w3,   \
                                        const float* w4, float* out, const int32_t* tpos, float* out_t,          \
                                        int64_t E, hipStream_t st) {                                             \
        const int64_t ntiles = gml_cdiv(E, 16);                                                                  \
        int64_t grid = gml_cdiv(ntiles, 4);                                                                      \
        if (grid > 4 * GML_NUM_CU) grid = 4 * GML_NUM_CU;   /* 112 VGPRs */                                      \
        hipLaunchKernelGGL((gml_k_edge_chain16_fwd<SV>), dim3((unsigned)grid), dim3(256), 0, st, es, w1, w2,     \
                           w3, w4, out, tpos, out_t, E, ntiles);                                                 \
        return gml_launch_status();                                                                              \
    }                                                                                                            \
    template <>                                                                                                  \
    int gml_launch_edge_chain16_bwd<SV>(const uint32_t* es, const float* w1, const float* w2, const float* w3,   \
                                        const float* w4, const float* gout, float* dw1, float* dw2, float* dw3,  \
                                        float* dw4, int64_t E, void* ws, size_t ws_bytes, hipStream_t st) {      \
        const int64_t ntiles = gml_cdiv(E, 16);                                                                  \
        const int64_t grid = gml_edge_chain16_bwd_groups(E);                                                     \
        constexpr int NW = GML_CHAIN16_NW(SV);                                                                   \
        if (ws_bytes < (size_t)grid * NW * sizeof(float)) return GML_E_WORKSPACE;                                \
        hipLaunchKernelGGL((gml_k_edge_chain16_bwd<SV>), dim3((unsigned)grid), dim3(256), 0, st, es, w1, w2,     \
                           w3, w4, gout, (float*)ws, E, ntiles);                                                 \
        int rc = gml_launch_status();                                                                            \
        if (rc != GML_OK) return rc;                                                                             \
        if (!dw1) return GML_OK;   /* partials stay in ws: gml_fold_many */                                      \
        const int n123 = 2 * SV * SV, n4 = SV * 4 * SV;                                                          \
        hipLaunchKernelGGL(gml_k_reduce_partials, dim3((unsigned)gml_cdiv(NW, 16)), dim3(256), 0, st,            \
                           (const float*)ws, grid, NW, dw1, n123, dw2, n123, dw3, n123, dw4, n4);                \
        return gml_launch_status();                                                                              \
    }
