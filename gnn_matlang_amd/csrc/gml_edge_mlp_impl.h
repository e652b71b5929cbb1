// ML3Layer edge branch for gfx950 (reference: /root/reference/libs/spect_conv.py:190-194,205-207)
//
//   out = relu( W4 . [ relu(W1 e) ; tanh(W2 e) * tanh(W3 e) ] )    e in R^S per edge, all bias-free
//
// One edge per lane.  The 10*S*S weights are wave-uniform and read through the scalar cache
// (s_load) as FMA scalar operands, so the forward is pure VALU on 32-byte coalesced records.
// Backward recomputes the intermediates from e (nothing but `out` was ever written), forms the
// per-edge gradients in registers and hands the four weight-gradient outer-product sums to the
// matrix cores: the wave transposes its 64 edges through a private LDS tile and contracts over the
// edge axis with v_mfma_f32_16x16x4_f32, accumulating across all its batches in registers; one
// deterministic partial per wave goes to the workspace and a second kernel folds them.
#pragma once
#include "gml_common.h"

// The weights are loop-invariant, so LICM would hoist all 10*S*S scalar loads out of the edge loop
// and spill hundreds of SGPRs; re-deriving the (uniform) pointers per iteration keeps every s_load
// next to its single use.
typedef const __attribute__((address_space(4))) float* gml_cptr;   // constant AS: uniform loads -> s_load
#define GML_LAUNDER_WEIGHTS()                                                          \
    gml_cptr w1, w2, w3, w4;                                                           \
    {                                                                                  \
        uint64_t a1 = (uint64_t)w1_, a2 = (uint64_t)w2_, a3 = (uint64_t)w3_, a4 = (uint64_t)w4_; \
        asm volatile("" : "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4));                     \
        w1 = (gml_cptr)a1; w2 = (gml_cptr)a2; w3 = (gml_cptr)a3; w4 = (gml_cptr)a4;    \
    }

template <int S, int SO>
struct GmlEdgeMlp {
    static constexpr int H2 = 2 * S, H4 = 4 * S;
    static constexpr int ROW_ALIGN = (S % 4 == 0) ? 4 : ((S % 2 == 0) ? 2 : 1);
    static constexpr int OUT_ALIGN = (SO % 4 == 0) ? 4 : ((SO % 2 == 0) ? 2 : 1);

    // z1 = W1 e, z2 = W2 e, z3 = W3 e ; h = [relu(z1); tanh(z2)*tanh(z3)]
    __device__ static __forceinline__ void hidden(const float (&e)[S], gml_cptr w1, gml_cptr w2, gml_cptr w3,
                                                  float (&z1)[H2], float (&t2)[H2], float (&t3)[H2]) {
#pragma unroll
        for (int o = 0; o < H2; ++o) {
            float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                a = fmaf(w1[o * S + i], e[i], a);
                b = fmaf(w2[o * S + i], e[i], b);
                c = fmaf(w3[o * S + i], e[i], c);
            }
            // (this family IS the exact-arithmetic road -- GML_F32_MFMA / S = 1 / d supports at S > 8: the library's tanh, <= 2 ulp,
            //  not gml_tanh's ~2e-7 absolute: with it the learned supports carried 5e-7 rms where torch's fp32 carries 5e-9)
            z1[o] = a; t2[o] = tanhf(b); t3[o] = tanhf(c);
        }
    }
};

template <int S, int SO>
__global__ __launch_bounds__(256) void gml_k_edge_mlp_fwd(const float* __restrict__ ea, const float* __restrict__ w1_,
                                                         const float* __restrict__ w2_, const float* __restrict__ w3_,
                                                         const float* __restrict__ w4_, float* __restrict__ out,
                                                         const int32_t* __restrict__ tpos, float* __restrict__ out_t,
                                                         int64_t E) {
    using M = GmlEdgeMlp<S, SO>;
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < E; e0 += (int64_t)gridDim.x * blockDim.x) {
        GML_LAUNDER_WEIGHTS();
        float e[S];
        gml_load_row<S, M::ROW_ALIGN>(ea + e0 * S, e);
        float z1[M::H2], t2[M::H2], t3[M::H2];
        M::hidden(e, w1, w2, w3, z1, t2, t3);
        float o[SO];
#pragma unroll
        for (int q = 0; q < SO; ++q) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < M::H2; ++c) {
                a = fmaf(w4[q * M::H4 + c], fmaxf(z1[c], 0.f), a);
                a = fmaf(w4[q * M::H4 + M::H2 + c], t2[c] * t3[c], a);
            }
            o[q] = fmaxf(a, 0.f);
        }
        // the same row goes out twice when the caller wants it: in this (target-sorted) order for the forward
        // kernel and at tpos[e] (source-sorted order) for the backward kernel, which then needs no gather pass
#pragma unroll
        for (int copy = 0; copy < 2; ++copy) {
            if (copy == 1 && out_t == nullptr) break;
            float* dst = (copy == 0) ? out + e0 * SO : out_t + (int64_t)tpos[e0] * SO;
            if constexpr (M::OUT_ALIGN == 4) {
#pragma unroll
                for (int q = 0; q < SO / 4; ++q)
                    *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
            } else {
#pragma unroll
                for (int q = 0; q < SO; ++q) dst[q] = o[q];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward.  Per-wave LDS tile T[64 edges][STR]; two stages reuse it:
//   stage A channels: go (SO) | h (4S)           -> dW4[SO, 4S]   = go^T h
//   stage B channels: gz1|gz2|gz3 (6S) | e (S)   -> dW1..3[2S, S] = gz^T e
// ---------------------------------------------------------------------------------------------
template <int S, int SO>
struct GmlEdgeMlpBwdCfg {
    static constexpr int CHA = SO + 4 * S, CHB = 7 * S;
    static constexpr int CH = (CHA > CHB ? CHA : CHB);
    static constexpr int STR = CH | 1;                     // odd stride: conflict-free column writes
    static constexpr int NB4 = (4 * S + 15) / 16;          // dW4 column blocks
    static constexpr int NBZ = (6 * S + 15) / 16;          // dW1..3 row blocks
    static constexpr int NW = 3 * 2 * S * S + SO * 4 * S;  // floats per partial
    static constexpr int WAVES = (STR * 64 * 4 * 4 <= 64 * 1024) ? 4 : 2;   // waves per workgroup
};

template <int S, int SO>
__global__ __launch_bounds__((GmlEdgeMlpBwdCfg<S, SO>::WAVES * 64))
void gml_k_edge_mlp_bwd(const float* __restrict__ ea, const float* __restrict__ w1_, const float* __restrict__ w2_,
                        const float* __restrict__ w3_, const float* __restrict__ w4_, const float* __restrict__ gout,
                        float* __restrict__ gin, float* __restrict__ partial, int64_t E, int64_t nbatch,
                        int64_t batches_per_wave) {
    using M = GmlEdgeMlp<S, SO>;
    using C = GmlEdgeMlpBwdCfg<S, SO>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* T = lds + wave * (64 * C::STR);
    const int i16 = lane & 15, kq = lane >> 4;
    const int64_t gwave = (int64_t)blockIdx.x * C::WAVES + wave;

    f32x4 acc4[C::NB4], accz[C::NBZ];
#pragma unroll
    for (int b = 0; b < C::NB4; ++b) acc4[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < C::NBZ; ++b) accz[b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int64_t b0 = gwave * batches_per_wave;
    const int64_t b1 = min(b0 + batches_per_wave, nbatch);
    for (int64_t b = b0; b < b1; ++b) {
        GML_LAUNDER_WEIGHTS();
        const int64_t e0 = b * 64 + lane;
        const bool valid = e0 < E;
        float e[S];
#pragma unroll
        for (int i = 0; i < S; ++i) e[i] = 0.f;
        if (valid) gml_load_row<S, M::ROW_ALIGN>(ea + e0 * S, e);
        float z1[M::H2], t2[M::H2], t3[M::H2];
        M::hidden(e, w1, w2, w3, z1, t2, t3);
        // go = gout * (o > 0), o recomputed
        float go[SO];
        {
            float g[SO];
#pragma unroll
            for (int q = 0; q < SO; ++q) g[q] = 0.f;
            if (valid) gml_load_row<SO, M::OUT_ALIGN>(gout + e0 * SO, g);
#pragma unroll
            for (int q = 0; q < SO; ++q) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < M::H2; ++c) {
                    a = fmaf(w4[q * M::H4 + c], fmaxf(z1[c], 0.f), a);
                    a = fmaf(w4[q * M::H4 + M::H2 + c], t2[c] * t3[c], a);
                }
                go[q] = (a > 0.f) ? g[q] : 0.f;
            }
        }
        // ---- stage A tile: go | h
#pragma unroll
        for (int q = 0; q < SO; ++q) T[lane * C::STR + q] = go[q];
#pragma unroll
        for (int c = 0; c < M::H2; ++c) {
            T[lane * C::STR + SO + c] = fmaxf(z1[c], 0.f);
            T[lane * C::STR + SO + M::H2 + c] = t2[c] * t3[c];
        }
        // per-edge hidden gradients (registers)
        float gz1[M::H2], gz2[M::H2], gz3[M::H2];
#pragma unroll
        for (int c = 0; c < M::H2; ++c) {
            float ga = 0.f, gb = 0.f;
#pragma unroll
            for (int q = 0; q < SO; ++q) {
                ga = fmaf(w4[q * M::H4 + c], go[q], ga);
                gb = fmaf(w4[q * M::H4 + M::H2 + c], go[q], gb);
            }
            gz1[c] = (z1[c] > 0.f) ? ga : 0.f;
            gz2[c] = gb * t3[c] * (1.f - t2[c] * t2[c]);
            gz3[c] = gb * t2[c] * (1.f - t3[c] * t3[c]);
        }
        if (gin != nullptr && valid) {
            float gi[S];
#pragma unroll
            for (int i = 0; i < S; ++i) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < M::H2; ++c) {
                    a = fmaf(w1[c * S + i], gz1[c], a);
                    a = fmaf(w2[c * S + i], gz2[c], a);
                    a = fmaf(w3[c * S + i], gz3[c], a);
                }
                gi[i] = a;
            }
#pragma unroll
            for (int i = 0; i < S; ++i) gin[e0 * S + i] = gi[i];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's tile writes have landed
        // dW4[q][c] += sum_e go[e][q] * h[e][c] :  A[i=q][k=e], B[k=e][j=c]
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int er = 4 * t + kq;
            const float a = (i16 < SO) ? T[er * C::STR + i16] : 0.f;
#pragma unroll
            for (int jb = 0; jb < C::NB4; ++jb) {
                const int c = jb * 16 + i16;
                const float bv = (c < M::H4) ? T[er * C::STR + SO + c] : 0.f;
                acc4[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc4[jb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // reads done before the tile is overwritten
        // ---- stage B tile: gz1 | gz2 | gz3 | e
#pragma unroll
        for (int c = 0; c < M::H2; ++c) {
            T[lane * C::STR + c] = gz1[c];
            T[lane * C::STR + M::H2 + c] = gz2[c];
            T[lane * C::STR + 2 * M::H2 + c] = gz3[c];
        }
#pragma unroll
        for (int i = 0; i < S; ++i) T[lane * C::STR + 6 * S + i] = e[i];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // dW123[z][i] += sum_e gz[e][z] * e[e][i] :  A[i=z][k=e], B[k=e][j=i]
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int er = 4 * t + kq;
            const float bv = (i16 < S) ? T[er * C::STR + 6 * S + i16] : 0.f;
#pragma unroll
            for (int ib = 0; ib < C::NBZ; ++ib) {
                const int z = ib * 16 + i16;
                const float a = (z < 6 * S) ? T[er * C::STR + z] : 0.f;
                accz[ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, accz[ib], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }

    // one partial per wave: [dw1 (2S*S) | dw2 | dw3 | dw4 (SO*4S)]; D[row = 4*kq + reg][col = i16]
    float* P = partial + gwave * C::NW;
#pragma unroll
    for (int ib = 0; ib < C::NBZ; ++ib)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int z = ib * 16 + 4 * kq + reg;      // row of the stacked [6S, S] gradient
            if (z < 6 * S && i16 < S) P[z * S + i16] = accz[ib][reg];
        }
#pragma unroll
    for (int jb = 0; jb < C::NB4; ++jb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int q = 4 * kq + reg, c = jb * 16 + i16;
            if (q < SO && c < M::H4) P[6 * S * S + q * M::H4 + c] = acc4[jb][reg];
        }
}

// dw[j] = sum_w partial[w][j]  (fixed order: deterministic)
__global__ void gml_k_reduce_partials(const float* __restrict__ partial, int64_t nwaves, int nw,
                                      float* __restrict__ d0, int n0, float* __restrict__ d1, int n1,
                                      float* __restrict__ d2, int n2, float* __restrict__ d3, int n3);

template <int S, int SO>
int gml_launch_edge_mlp_fwd(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4,
                            float* out, const int32_t* tpos, float* out_t, int64_t E, hipStream_t st);
template <int S, int SO>
int gml_launch_edge_mlp_bwd(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4,
                            const float* gout, float* gin, float* dw1, float* dw2, float* dw3, float* dw4,
                            int64_t E, void* ws, size_t ws_bytes, hipStream_t st);

static inline int64_t gml_edge_mlp_bwd_waves(int64_t E, int waves_per_wg) {
    const int64_t nbatch = gml_cdiv(E, 64);
    int64_t nw = (int64_t)GML_NUM_CU * 8;            // persistent: <= 8 waves per CU hold accumulators
    if (nw > nbatch) nw = nbatch;
    nw = gml_cdiv(nw, waves_per_wg) * waves_per_wg;
    return nw < waves_per_wg ? waves_per_wg : nw;
}

#define GML_DEFINE_EDGE_MLP(SV)                                                                              \
    template <>                                                                                              \
    int gml_launch_edge_mlp_fwd<SV, SV>(const float* ea, const float* w1, const float* w2, const float* w3,  \
                                        const float* w4, float* out, const int32_t* tpos, float* out_t,      \
                                        int64_t E, hipStream_t st) {                                         \
        int64_t grid = gml_cdiv(E, 256);                                                                     \
        if (grid > GML_NUM_CU * 16) grid = GML_NUM_CU * 16;                                                  \
        hipLaunchKernelGGL((gml_k_edge_mlp_fwd<SV, SV>), dim3((unsigned)grid), dim3(256), 0, st, ea, w1, w2, \
                           w3, w4, out, tpos, out_t, E);                                                     \
        return gml_launch_status();                                                                          \
    }                                                                                                        \
    template <>                                                                                              \
    int gml_launch_edge_mlp_bwd<SV, SV>(const float* ea, const float* w1, const float* w2, const float* w3,  \
                                        const float* w4, const float* gout, float* gin, float* dw1,          \
                                        float* dw2, float* dw3, float* dw4, int64_t E, void* ws,             \
                                        size_t ws_bytes, hipStream_t st) {                                   \
        using C = GmlEdgeMlpBwdCfg<SV, SV>;                                                                  \
        const int64_t nbatch = gml_cdiv(E, 64);                                                              \
        const int64_t nwaves = gml_edge_mlp_bwd_waves(E, C::WAVES);                                          \
        if (ws_bytes < (size_t)nwaves * C::NW * sizeof(float)) return GML_E_WORKSPACE;                       \
        const int64_t bpw = gml_cdiv(nbatch, nwaves);                                                        \
        const size_t lds = (size_t)C::WAVES * 64 * C::STR * sizeof(float);                                   \
        hipLaunchKernelGGL((gml_k_edge_mlp_bwd<SV, SV>), dim3((unsigned)(nwaves / C::WAVES)),                \
                           dim3(C::WAVES * 64), lds, st, ea, w1, w2, w3, w4, gout, gin, (float*)ws, E,       \
                           nbatch, bpw);                                                                     \
        int rc = gml_launch_status();                                                                        \
        if (rc != GML_OK) return rc;                                                                         \
        if (!dw1) return GML_OK;   /* partials stay in ws: gml_fold_many */                                  \
        const int n123 = 2 * SV * SV, n4 = SV * 4 * SV;                                                      \
        hipLaunchKernelGGL(gml_k_reduce_partials, dim3((unsigned)gml_cdiv(C::NW, 16)), dim3(256), 0, st,      \
                           (const float*)ws, nwaves, C::NW, dw1, n123, dw2, n123, dw3, n123, dw4, n4);       \
        return gml_launch_status();                                                                          \
    }
