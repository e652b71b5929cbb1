// ML3Layer edge branch FORWARD with fp32-class products for 8 < S = Sout <= 16 (counting.py: S = 12; reference:
// /root/reference/libs/spect_conv.py:190-194, 205-207): the three-piece arithmetic of gml_edge_chain6_impl.h on the machine mapping of
// gml_edge_chain16_impl.h (tiles of 16 edges, edge = column; 2 S <= 32 hidden rows = two row tiles; 16 in-channels = 16 of the 32
// k-slots of an instruction, so an instruction carries TWO piece products):
//
//   layer 1   three instructions per (weight matrix, row tile):  [Wh | Wh] . [e_h | e_m]  +  [Wm | Wm] . [e_h | e_m]  +  [Wh | Wl] . [e_l | e_h]
//             = the six products down to 2^-24 (h h, h m, m h, m m, h l, l h)
//   layer 2   W4 [S x 4S], K = 4S <= 64 = two K = 32 steps (the h1 tile pair, the h23 tile pair) x six piece products
//
// The fp32 rows of the supports are read directly (64 bytes per edge at S = 16: what the pre-split rows of the two-piece chain cost) and
// cut in registers; the residual subtractions of the hidden activations run on the matrix pipe (gml_split3_tiles).
#pragma once
#include "gml_edge_chain6_impl.h"

template <int S>
struct GmlChain16W6 {
    bf16x8 a1[3][2][3];       // [W1, W2, W3][row tile][instruction]
    bf16x8 a2[2][3];          // W4: [k-step: h1 tiles / h23 tiles][piece h, m, l]
};

template <int S>
__device__ __forceinline__ void gml_chain16x6_load_weights(GmlChain16W6<S>& W, const float* __restrict__ w1, const float* __restrict__ w2,
                                                           const float* __restrict__ w3, const float* __restrict__ w4, int c16, int g) {
    constexpr int H2 = 2 * S, H4 = 4 * S;
    const float* w123[3] = {w1, w2, w3};
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
            const GmlOp3 p = gml_wop3(v);
            W.a1[b][t][0] = p.h;
            W.a1[b][t][1] = p.m;
            W.a1[b][t][2] = g < 2 ? p.h : p.l;
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
        const GmlOp3 p = gml_wop3(v);
        W.a2[st][0] = p.h; W.a2[st][1] = p.m; W.a2[st][2] = p.l;
    }
}

// out [E, S]; out_t (optional): the same rows at tpos[e].  SYM (gml_edge_chain_sym_impl.h): tile entry u evaluates edge uid[u] and
// stores the row to out[uid[u]] and out[mir[u]] (mir < 0: none); E is then the number of entries (tpos / out_t unused)
template <int S, bool TA, bool SYM = false>
__global__ __launch_bounds__(256, 2) void gml_k_edge_chain16x6_fwd(const float* __restrict__ ea, const float* __restrict__ w1,
                                                                  const float* __restrict__ w2, const float* __restrict__ w3,
                                                                  const float* __restrict__ w4, float* __restrict__ out,
                                                                  const int32_t* __restrict__ tpos, float* __restrict__ out_t,
                                                                  int64_t E, int64_t ntiles, const int32_t* __restrict__ uid = nullptr,
                                                                  const int32_t* __restrict__ mir = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    GmlChain16W6<S> W;
    gml_chain16x6_load_weights<S>(W, w1, w2, w3, w4, c16, g);
    GmlNegI negI;
    gml_chain_make_negI(negI, c16, g);
    const int q0 = 4 * g, c0 = 8 * (g & 1);                    // output rows / in-channels of this lane group
    // a wave works on PAIRS of tiles (two independent dependency chains per trip: the kernel is latency-bound, not issue-bound, with one)
    const int64_t stride = (int64_t)gridDim.x * 8;
    int64_t t = ((int64_t)blockIdx.x * 4 + wave) * 2;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    struct Tile { float e[8]; int32_t tp, sid; };
    auto fetch = [&](int64_t tt, Tile& T) {                     // clamped: always a readable edge
        int64_t ed = min(tt * 16 + c16, E - 1);
        T.sid = 0;
        if constexpr (SYM) {
            T.tp = mir[ed];
            ed = T.sid = uid[ed];
        }
        const float* p = ea + ed * S;
        if constexpr (S % 4 == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = c0 + 4 * i;
                const f32x4 v = c < S ? *reinterpret_cast<const f32x4*>(p + c) : zero;
                T.e[4 * i] = v.x; T.e[4 * i + 1] = v.y; T.e[4 * i + 2] = v.z; T.e[4 * i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) T.e[j] = c0 + j < S ? p[c0 + j] : 0.f;
        }
        if constexpr (!SYM) T.tp = out_t != nullptr ? tpos[ed] : 0;
    };
    auto chain = [&](const Tile& T) -> f32x4 {
        // layer-1 operands: BA = [e_h | e_m], BB = [e_l | e_h] over the lane groups (0, 1 | 2, 3)
        uint32_t a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t h, m, l;
            gml_split3_pair(T.e[2 * j], T.e[2 * j + 1], h, m, l);
            a[j] = g < 2 ? h : m;
            b[j] = g < 2 ? l : h;
        }
        const bf16x8 BA = gml_op(a[0], a[1], a[2], a[3]), BB = gml_op(b[0], b[1], b[2], b[3]);
        f32x4 h1[2], h23[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4 z[3];
#pragma unroll
            for (int mtx = 0; mtx < 3; ++mtx)
                z[mtx] = GML_MFMA(W.a1[mtx][rt][0], BA, GML_MFMA(W.a1[mtx][rt][1], BA, GML_MFMA(W.a1[mtx][rt][2], BB, zero)));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t2, t3;
                gml_tanh_pair_scaled<TA>(z[1][r], z[2][r], t2, t3);
                h1[rt][r] = gml_relu1(z[0][r]);
                h23[rt][r] = t2 * t3;
            }
        }
        u32x4 p1h, p1m, p1l, p2h, p2m, p2l;
        gml_split3_tiles(negI, h1[0], h1[1], p1h, p1m, p1l);
        gml_split3_tiles(negI, h23[0], h23[1], p2h, p2m, p2l);
        const bf16x8 A1[3] = {__builtin_bit_cast(bf16x8, p1h), __builtin_bit_cast(bf16x8, p1m), __builtin_bit_cast(bf16x8, p1l)};
        const bf16x8 A2[3] = {__builtin_bit_cast(bf16x8, p2h), __builtin_bit_cast(bf16x8, p2m), __builtin_bit_cast(bf16x8, p2l)};
        // two independent accumulation chains (the h1 step, the h23 step), small products first
        f32x4 o = GML_MFMA(W.a2[0][2], A1[0], zero), o2 = GML_MFMA(W.a2[1][2], A2[0], zero);
        o = GML_MFMA(W.a2[0][1], A1[1], o);  o2 = GML_MFMA(W.a2[1][1], A2[1], o2);
        o = GML_MFMA(W.a2[0][0], A1[2], o);  o2 = GML_MFMA(W.a2[1][0], A2[2], o2);
        o = GML_MFMA(W.a2[0][1], A1[0], o);  o2 = GML_MFMA(W.a2[1][1], A2[0], o2);
        o = GML_MFMA(W.a2[0][0], A1[1], o);  o2 = GML_MFMA(W.a2[1][0], A2[1], o2);
        o = GML_MFMA(W.a2[0][0], A1[0], o);  o2 = GML_MFMA(W.a2[1][0], A2[0], o2);
        return o + o2;
    };
    auto store = [&](int64_t tt, const Tile& T, const f32x4 o) {
        const int64_t eid = tt * 16 + c16;
        if (tt < ntiles && eid < E && q0 < S) {
            const f32x4 v = f32x4{gml_relu1(o[0]), gml_relu1(o[1]), gml_relu1(o[2]), gml_relu1(o[3])};
            float* op = out + (SYM ? (int64_t)T.sid : eid) * S + q0;
            float* ot = SYM ? (T.tp >= 0 ? out + (int64_t)T.tp * S + q0 : nullptr) : (out_t != nullptr ? out_t + (int64_t)T.tp * S + q0 : nullptr);
            if constexpr (S % 4 == 0) {
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(op));
                if (ot != nullptr) *reinterpret_cast<f32x4*>(ot) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (q0 + r < S) {
                        op[r] = v[r];
                        if (ot != nullptr) ot[r] = v[r];
                    }
            }
        }
    };
    Tile N0, N1;
    fetch(t, N0);
    fetch(t + 1, N1);
    for (; t < ntiles; t += stride) {
        const Tile C0 = N0, C1 = N1;
        const int64_t tn = t + stride < ntiles ? t + stride : t;
        fetch(tn, N0);                                          // next pair's rows in flight during these chains
        fetch(tn + 1, N1);
        const f32x4 o0 = chain(C0), o1 = chain(C1);
        store(t, C0, o0);
        store(t + 1, C1, o1);
    }
}

template <int S>
int gml_launch_edge_chain16x6_fwd_sym(const float* ea, const int32_t* uid, const int32_t* mir, int64_t U, const float* w1, const float* w2,
                                      const float* w3, const float* w4, float* out, hipStream_t st) {
    const int64_t ntiles = gml_cdiv(U, 16);
    int64_t grid = gml_cdiv(ntiles, 8);
    if (grid > 2 * GML_NUM_CU) grid = 2 * GML_NUM_CU;
    if (gml_chain6_accurate_tanh())
        hipLaunchKernelGGL((gml_k_edge_chain16x6_fwd<S, true, true>), dim3((unsigned)grid), dim3(256), 0, st, ea, w1, w2, w3, w4, out, nullptr, nullptr, U, ntiles, uid, mir);
    else
        hipLaunchKernelGGL((gml_k_edge_chain16x6_fwd<S, false, true>), dim3((unsigned)grid), dim3(256), 0, st, ea, w1, w2, w3, w4, out, nullptr, nullptr, U, ntiles, uid, mir);
    return gml_launch_status();
}

template <int S>
int gml_launch_edge_chain16x6_fwd(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, float* out,
                                  const int32_t* tpos, float* out_t, int64_t E, hipStream_t st) {
    const int64_t ntiles = gml_cdiv(E, 16);
    int64_t grid = gml_cdiv(ntiles, 8);
    if (grid > 2 * GML_NUM_CU) grid = 2 * GML_NUM_CU;
    if (gml_chain6_accurate_tanh())
        hipLaunchKernelGGL((gml_k_edge_chain16x6_fwd<S, true>), dim3((unsigned)grid), dim3(256), 0, st, ea, w1, w2, w3, w4, out, tpos, out_t, E, ntiles);
    else
        hipLaunchKernelGGL((gml_k_edge_chain16x6_fwd<S, false>), dim3((unsigned)grid), dim3(256), 0, st, ea, w1, w2, w3, w4, out, tpos, out_t, E, ntiles);
    return gml_launch_status();
}
