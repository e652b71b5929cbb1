// ML3Layer edge branch for MORE than 16 supports (16 < max(S, Sout) <= 48), forward -- SURVEY s8(d) workload R raises the support
// count of the sr25 sweep to 24 and 48; reference: /root/reference/libs/spect_conv.py:190-194 (fc1_1..fc1_4, bias-free), :205-207
//
//   out = relu( W4 . [ relu(W1 e) ; tanh(W2 e) * tanh(W3 e) ] )          e in R^S per edge, out in R^Sout
//
// Rounds 1-4 ran this case as four library GEMMs + elementwise kernels under autograd: every intermediate ([E, 2S] three times,
// their activations, the product) went through HBM -- 46 / 84 ms forward at S = 24 / 48 on 13 M edges.  Here one launch reads 4 S
// bytes and writes 4 Sout bytes per edge; nothing else leaves the CU.
//
// One edge per lane, exact fp32 FMAs (no split products: the oracle's arithmetic up to summation order).  The 10 S^2 weights live in
// LDS, zero padded to SP = roundup4(max(S, Sout)): W1..W3 as [matrix][hidden unit][SP inputs], W4 transposed as [hidden unit][SP
// outputs].  A workgroup walks the hidden layer four units at a time: 12 accumulator chains over the lane's e registers (every
// weight quad = one broadcast ds_read_b128 feeding 4 FMAs), activations, then the four units' rank-1 updates of the SP output
// accumulators (again one broadcast quad per 4 FMAs).  The kernel is bound by those broadcast reads (one LDS cycle group per
// 4 FMAs of a wave): ~8 ms at S = 48, ~2 ms at S = 24 on 13 M edges.
#include "gml_common.h"

template <int SP>
__global__ __launch_bounds__(512) void gml_k_edge_wide_fwd(const float* __restrict__ ea, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, const float* __restrict__ w3,
                                                          const float* __restrict__ w4, float* __restrict__ out, int64_t E,
                                                          int S, int So) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int H2 = 2 * S, H2R = (H2 + 3) & ~3;               // hidden units per branch; chunks of 4 actually walked
    constexpr int H2P = 2 * SP;                              // rows of the images: a compile-time stride (every LDS offset an immediate)
    float* W123 = wl;                                        // [3][SP inputs][H2P hidden units]: a quad = 4 consecutive units at one input
    float* W4T = wl + 3 * H2P * SP;                          // [2][H2P][SP]: rows of branch 0 (relu units), then branch 1 (tanh products)
    for (int i = threadIdx.x; i < 3 * H2P * SP; i += blockDim.x) {
        const int u = i % H2P, j = (i / H2P) % SP, m = i / (SP * H2P);
        const float* w = m == 0 ? w1 : (m == 1 ? w2 : w3);
        // W2, W3 carry the 2 log2(e) of tanh(z) = 1 - 2 / (2^(2 log2(e) z) + 1)
        W123[i] = (u < H2 && j < S) ? w[u * S + j] * (m == 0 ? 1.f : 2.8853900817779268f) : 0.f;
    }
    for (int i = threadIdx.x; i < 2 * H2P * SP; i += blockDim.x) {
        const int q = i % SP, u = (i / SP) % H2P, br = i / (SP * H2P);
        W4T[i] = (u < H2 && q < So) ? w4[q * (4 * S) + br * H2 + u] : 0.f;
    }
    __syncthreads();
    const bool vec_in = (S % 4 == 0), vec_out = (So % 4 == 0);
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < E; e0 += (int64_t)gridDim.x * blockDim.x) {
        float e[SP];
        const float* er = ea + e0 * S;
        if (vec_in) {
#pragma unroll
            for (int j = 0; j < SP / 4; ++j) {
                const f32x4 t = (4 * j < S) ? *reinterpret_cast<const f32x4*>(er + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
                e[4 * j] = t.x; e[4 * j + 1] = t.y; e[4 * j + 2] = t.z; e[4 * j + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SP; ++j) e[j] = j < S ? er[j] : 0.f;
        }
        f32x2 o2[SP / 2];
#pragma unroll
        for (int q = 0; q < SP / 2; ++q) o2[q] = f32x2{0.f, 0.f};
        for (int u0 = 0; u0 < H2R; u0 += 4) {
            const float* Wa = W123 + u0;                     // moving bases: the offsets below are compile-time constants
            const float* Wb = W4T + u0 * SP;
            f32x2 zz[3][2];                                  // (units u0, u0 + 1), (u0 + 2, u0 + 3): packed FMAs with e[j] broadcast
#pragma unroll
            for (int m = 0; m < 3; ++m) { zz[m][0] = f32x2{0.f, 0.f}; zz[m][1] = f32x2{0.f, 0.f}; }
            // weights in batches of 12 quads (4 inputs x 3 matrices) ahead of their 24 packed FMAs: left to itself the scheduler
            // keeps ONE quad in flight (read, wait, two FMAs: the LDS latency of every read exposed)
#pragma unroll
            for (int j4 = 0; j4 < SP / 4; ++j4) {
                f32x4 wq[4][3];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int m = 0; m < 3; ++m) wq[jj][m] = *reinterpret_cast<const f32x4*>(Wa + (m * SP + 4 * j4 + jj) * H2P);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f32x2 ep = f32x2{e[4 * j4 + (jj & 2)], e[4 * j4 + (jj & 2) + 1]};   // a register pair: its halves broadcast through op_sel
                    const f32x2 ej = (jj & 1) ? __builtin_shufflevector(ep, ep, 1, 1) : __builtin_shufflevector(ep, ep, 0, 0);
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        zz[m][0] = f32x2{wq[jj][m].x, wq[jj][m].y} * ej + zz[m][0];
                        zz[m][1] = f32x2{wq[jj][m].z, wq[jj][m].w} * ej + zz[m][1];
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
            }
            float z[3][4];
#pragma unroll
            for (int m = 0; m < 3; ++m) { z[m][0] = zz[m][0].x; z[m][1] = zz[m][0].y; z[m][2] = zz[m][1].x; z[m][3] = zz[m][1].y; }
            float h[2][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                h[0][u] = fmaxf(z[0][u], 0.f);
                const float t2 = fmaf(-2.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z[1][u]) + 1.f), 1.f);
                const float t3 = fmaf(-2.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z[2][u]) + 1.f), 1.f);
                h[1][u] = t2 * t3;
            }
#pragma unroll
            for (int br = 0; br < 2; ++br)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f32x4 wq[SP / 4];
#pragma unroll
                    for (int q = 0; q < SP / 4; ++q) wq[q] = *reinterpret_cast<const f32x4*>(Wb + (br * H2P + u) * SP + 4 * q);
                    const f32x2 hb = f32x2{h[br][u], h[br][u]};
#pragma unroll
                    for (int q = 0; q < SP / 4; ++q) {
                        o2[2 * q] = f32x2{wq[q].x, wq[q].y} * hb + o2[2 * q];
                        o2[2 * q + 1] = f32x2{wq[q].z, wq[q].w} * hb + o2[2 * q + 1];
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, SP / 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, SP / 2, 0);
                }
        }
        float o[SP];
#pragma unroll
        for (int q = 0; q < SP / 2; ++q) { o[2 * q] = o2[q].x; o[2 * q + 1] = o2[q].y; }
        float* dst = out + e0 * So;
        if (vec_out) {
#pragma unroll
            for (int q = 0; q < SP / 4; ++q)
                if (4 * q < So)
                    *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{fmaxf(o[4 * q], 0.f), fmaxf(o[4 * q + 1], 0.f), fmaxf(o[4 * q + 2], 0.f), fmaxf(o[4 * q + 3], 0.f)};
        } else {
#pragma unroll
            for (int q = 0; q < SP; ++q)
                if (q < So) dst[q] = fmaxf(o[q], 0.f);
        }
    }
}

template <int SP>
static int launch_wide(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, float* out, int64_t E,
                       int S, int So, hipStream_t st) {
    const size_t lds = (size_t)5 * (2 * SP) * SP * sizeof(float);
    GML_ALLOW_BIG_LDS(rc, (&gml_k_edge_wide_fwd<SP>), 160 * 1024)
    if (rc != hipSuccess) return (int)rc;
    int64_t grid = gml_cdiv(E, 512);
    if (grid > GML_NUM_CU) grid = GML_NUM_CU;                 // one persistent workgroup per CU (its LDS image is 50-92 KB)
    hipLaunchKernelGGL((gml_k_edge_wide_fwd<SP>), dim3((unsigned)grid), dim3(512), lds, st, ea, w1, w2, w3, w4, out, E, S, So);
    return gml_launch_status();
}

// out[e, :] = relu(W4 [relu(W1 ea[e]); tanh(W2 ea[e]) * tanh(W3 ea[e])]) for 16 < max(S, Sout) <= 48 (smaller shapes: gml_edge_mlp_fwd).
// ea [E, S], w1..w3 [2S, S], w4 [Sout, 4S] row-major fp32, out [E, Sout]; ea / out rows 16-byte aligned when S / Sout are multiples of 4.
extern "C" int gml_edge_mlp_wide_fwd(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, float* out,
                                     int64_t E, int32_t S, int32_t Sout, gml_stream_t stream) {
    if (E < 0 || S <= 0 || Sout <= 0) return GML_E_BADARG;
    const int m = S > Sout ? S : Sout;
    if (m <= 16 || m > 48) return GML_E_UNSUPPORTED;
    if (E == 0) return GML_OK;
    if (!ea || !w1 || !w2 || !w3 || !w4 || !out) return GML_E_BADARG;
    if ((S % 4 == 0 && ((uintptr_t)ea & 15)) || (Sout % 4 == 0 && ((uintptr_t)out & 15))) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int sp = (m + 3) & ~3;
    switch (sp) {
        case 20: return launch_wide<20>(ea, w1, w2, w3, w4, out, E, S, Sout, st);
        case 24: return launch_wide<24>(ea, w1, w2, w3, w4, out, E, S, Sout, st);
        case 28: return launch_wide<28>(ea, w1, w2, w3, w4, out, E, S, Sout, st);
        case 32: return launch_wide<32>(ea, w1, w2, w3, w4, out, E, S, Sout, st);
        case 36: return launch_wide<36>(ea, w1, w2, w3, w4, out, E, S, Sout, st);
        case 40: return launch_wide<40>(ea, w1, w2, w3, w4, out, E, S, Sout, st);
        case 44: return launch_wide<44>(ea, w1, w2, w3, w4, out, E, S, Sout, st);
        case 48: return launch_wide<48>(ea, w1, w2, w3, w4, out, E, S, Sout, st);
    }
    return GML_E_UNSUPPORTED;
}

// =============================================================================================
// Backward for 16 < max(S, Sout) <= 48 (round 6; VERDICT r05 "missing" #2; reference: the autograd of libs/spect_conv.py:205-207).
// Rounds 1-5 recomputed the library expression under autograd.  One launch, one edge per lane, the forward's LDS images: the lane
// reads its support row e, its output row (the saved forward result: the relu mask) and its output-gradient row, walks the hidden
// layer four units at a time -- the 12 pre-activation chains of the forward, then dh = W4^T go for those units (8 dot products against
// the lane's go registers: the same broadcast quads the forward's rank-1 updates read) -- and writes, per edge,
//
//   go [E, So]           = gout * (out > 0)
//   hid [E, 2 * H2R]     = relu(W1 e) | tanh(W2 e) * tanh(W3 e)                     (H2R = 2 S rounded up to a multiple of 4)
//   gz  [E, 3 * H2R]     = dL/d(W1 e) | dL/d(W2 e) | dL/d(W3 e)
//
// from which the four weight gradients are the tall contractions gz_m^T e and hid^T go (gml_xty_wide: bf16x3 on the matrix cores,
// rows along K).  The supports' own gradient, when wanted, is sum_m gz_m W_m (three library GEMMs on the host side: no reference
// script asks for it).  Exact fp32 products here (the derivative factors as 4 e r^2, gml_tanh_d's form).
template <int SP, bool VEC>
__global__ __launch_bounds__(512) void gml_k_edge_wide_bwd(const float* __restrict__ ea, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, const float* __restrict__ w3,
                                                          const float* __restrict__ w4, const float* __restrict__ out,
                                                          const float* __restrict__ gout, float* __restrict__ go_out,
                                                          float* __restrict__ hid, float* __restrict__ gz, int64_t E, int S, int So) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int H2 = 2 * S, H2R = (H2 + 3) & ~3;
    constexpr int H2P = 2 * SP;
    float* W123 = wl;                                        // [3][SP inputs][H2P hidden units] (W2, W3 scaled by 2 log2(e))
    float* W4T = wl + 3 * H2P * SP;                          // [2][H2P][SP outputs]
    for (int i = threadIdx.x; i < 3 * H2P * SP; i += blockDim.x) {
        const int u = i % H2P, j = (i / H2P) % SP, m = i / (SP * H2P);
        const float* w = m == 0 ? w1 : (m == 1 ? w2 : w3);
        W123[i] = (u < H2 && j < S) ? w[u * S + j] * (m == 0 ? 1.f : 2.8853900817779268f) : 0.f;
    }
    for (int i = threadIdx.x; i < 2 * H2P * SP; i += blockDim.x) {
        const int q = i % SP, u = (i / SP) % H2P, br = i / (SP * H2P);
        W4T[i] = (u < H2 && q < So) ? w4[q * (4 * S) + br * H2 + u] : 0.f;
    }
    __syncthreads();
    constexpr bool vec_in = VEC, vec_out = VEC;               // (rows of 4-float multiples; otherwise scalar accesses)
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < E; e0 += (int64_t)gridDim.x * blockDim.x) {
        float e[SP];
        f32x2 go2[SP / 2];
        {
            const float* er = ea + e0 * S;
            if constexpr (vec_in) {
#pragma unroll
                for (int j = 0; j < SP / 4; ++j) {
                    const f32x4 t = (4 * j < S) ? *reinterpret_cast<const f32x4*>(er + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
                    e[4 * j] = t.x; e[4 * j + 1] = t.y; e[4 * j + 2] = t.z; e[4 * j + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < SP; ++j) e[j] = j < S ? er[j] : 0.f;
            }
            const float* orow = out + e0 * So;
            const float* grow = gout + e0 * So;
            float* gdst = go_out + e0 * So;
            float gq[SP];
            if constexpr (vec_out) {
#pragma unroll
                for (int q = 0; q < SP / 4; ++q) {
                    f32x4 g4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (4 * q < So) {
                        const f32x4 o4 = *reinterpret_cast<const f32x4*>(orow + 4 * q);
                        g4 = *reinterpret_cast<const f32x4*>(grow + 4 * q);
                        g4.x = o4.x > 0.f ? g4.x : 0.f; g4.y = o4.y > 0.f ? g4.y : 0.f;
                        g4.z = o4.z > 0.f ? g4.z : 0.f; g4.w = o4.w > 0.f ? g4.w : 0.f;
                        *reinterpret_cast<f32x4*>(gdst + 4 * q) = g4;
                    }
                    gq[4 * q] = g4.x; gq[4 * q + 1] = g4.y; gq[4 * q + 2] = g4.z; gq[4 * q + 3] = g4.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < SP; ++q) {
                    gq[q] = (q < So && orow[q] > 0.f) ? grow[q] : 0.f;
                    if (q < So) gdst[q] = gq[q];
                }
            }
#pragma unroll
            for (int q = 0; q < SP / 2; ++q) go2[q] = f32x2{gq[2 * q], gq[2 * q + 1]};
        }
        float* hrow = hid + e0 * (2 * H2R);
        float* zrow = gz + e0 * (3 * H2R);
        for (int u0 = 0; u0 < H2R; u0 += 4) {
            const float* Wa = W123 + u0;
            const float* Wb = W4T + u0 * SP;
            f32x2 zz[3][2];
#pragma unroll
            for (int m = 0; m < 3; ++m) { zz[m][0] = f32x2{0.f, 0.f}; zz[m][1] = f32x2{0.f, 0.f}; }
#pragma unroll
            for (int j4 = 0; j4 < SP / 4; ++j4) {
                f32x4 wq[4][3];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int m = 0; m < 3; ++m) wq[jj][m] = *reinterpret_cast<const f32x4*>(Wa + (m * SP + 4 * j4 + jj) * H2P);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f32x2 ep = f32x2{e[4 * j4 + (jj & 2)], e[4 * j4 + (jj & 2) + 1]};
                    const f32x2 ej = (jj & 1) ? __builtin_shufflevector(ep, ep, 1, 1) : __builtin_shufflevector(ep, ep, 0, 0);
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        zz[m][0] = f32x2{wq[jj][m].x, wq[jj][m].y} * ej + zz[m][0];
                        zz[m][1] = f32x2{wq[jj][m].z, wq[jj][m].w} * ej + zz[m][1];
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
            }
            float z[3][4];
#pragma unroll
            for (int m = 0; m < 3; ++m) { z[m][0] = zz[m][0].x; z[m][1] = zz[m][0].y; z[m][2] = zz[m][1].x; z[m][3] = zz[m][1].y; }
            // dh[br][u] = sum_q W4[q][br * 2S + u] go[q]: one broadcast quad per two packed FMAs, the halves summed at the end
            float dh[2][4];
#pragma unroll
            for (int br = 0; br < 2; ++br) {
                f32x2 a2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a2[u] = f32x2{0.f, 0.f};
#pragma unroll
                for (int q = 0; q < SP / 4; ++q) {             // four units per output quad: 4 broadcast reads, 8 packed FMAs
                    f32x4 wq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) wq[u] = *reinterpret_cast<const f32x4*>(Wb + (br * H2P + u) * SP + 4 * q);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a2[u] = f32x2{wq[u].x, wq[u].y} * go2[2 * q] + a2[u];
                        a2[u] = f32x2{wq[u].z, wq[u].w} * go2[2 * q + 1] + a2[u];
                    }
                    // (the eight dot products of a chunk are independent: left alone the scheduler issues all 8 x SP / 4 quad reads first
                    //  -- 8 SP registers, 300 .. 900 spilled; a memory clobber every second step keeps two steps of reads in flight)
                    if (q & 1) asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) dh[br][u] = a2[u].x + a2[u].y;
            }
            f32x4 h1v, h23v, g1v, g2v, g3v;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // tanh and its derivative from e = exp(2 x), r = 1 / (e + 1): t = 1 - 2 r, 1 - t^2 = 4 e r^2 (no cancellation near saturation)
                const float e2 = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(z[1][u], -115.f, 115.f));
                const float e3 = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(z[2][u], -115.f, 115.f));
                const float r2 = __builtin_amdgcn_rcpf(e2 + 1.f), r3 = __builtin_amdgcn_rcpf(e3 + 1.f);
                const float t2 = fmaf(-2.f, r2, 1.f), t3 = fmaf(-2.f, r3, 1.f);
                const float d2 = 4.f * (e2 * r2) * r2, d3 = 4.f * (e3 * r3) * r3;
                h1v[u] = fmaxf(z[0][u], 0.f);
                h23v[u] = t2 * t3;
                g1v[u] = z[0][u] > 0.f ? dh[0][u] : 0.f;
                g2v[u] = dh[1][u] * t3 * d2;
                g3v[u] = dh[1][u] * t2 * d3;
            }
            *reinterpret_cast<f32x4*>(hrow + u0) = h1v;
            *reinterpret_cast<f32x4*>(hrow + H2R + u0) = h23v;
            *reinterpret_cast<f32x4*>(zrow + u0) = g1v;
            *reinterpret_cast<f32x4*>(zrow + H2R + u0) = g2v;
            *reinterpret_cast<f32x4*>(zrow + 2 * H2R + u0) = g3v;
        }
    }
}

template <int SP>
static int launch_wide_bwd(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, const float* out,
                           const float* gout, float* go, float* hid, float* gz, int64_t E, int S, int So, hipStream_t st) {
    const size_t lds = (size_t)5 * (2 * SP) * SP * sizeof(float);
    int64_t grid = gml_cdiv(E, 512);
    if (grid > GML_NUM_CU) grid = GML_NUM_CU;
    if (S % 4 == 0 && So % 4 == 0) {
        GML_ALLOW_BIG_LDS(rc, (&gml_k_edge_wide_bwd<SP, true>), 160 * 1024)
        if (rc != hipSuccess) return (int)rc;
        hipLaunchKernelGGL((gml_k_edge_wide_bwd<SP, true>), dim3((unsigned)grid), dim3(512), lds, st, ea, w1, w2, w3, w4, out, gout, go, hid, gz, E, S, So);
    } else {
        GML_ALLOW_BIG_LDS(rc, (&gml_k_edge_wide_bwd<SP, false>), 160 * 1024)
        if (rc != hipSuccess) return (int)rc;
        hipLaunchKernelGGL((gml_k_edge_wide_bwd<SP, false>), dim3((unsigned)grid), dim3(512), lds, st, ea, w1, w2, w3, w4, out, gout, go, hid, gz, E, S, So);
    }
    return gml_launch_status();
}

// per-edge part of the backward of gml_edge_mlp_wide_fwd.  out: the forward's result [E, Sout]; gout: dL/dout [E, Sout]; writes
// go [E, Sout], hid [E, 2 * H2R], gz [E, 3 * H2R] with H2R = (2 S + 3) & ~3 (gml_edge_mlp_wide_bwd_h2r) -- rows 16-byte aligned.
// The weight gradients are then dW_m = gz[:, m H2R : m H2R + 2 S]^T ea and dW4 = (hid[:, br H2R : br H2R + 2 S]^T go)^T per branch
// (gml_xty_wide).  GML_E_UNSUPPORTED for max(S, Sout) <= 16 (gml_edge_mlp_bwd) or > 48.
extern "C" int32_t gml_edge_mlp_wide_bwd_h2r(int32_t S) { return (2 * S + 3) & ~3; }
extern "C" int gml_edge_mlp_wide_bwd(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, const float* out,
                                     const float* gout, float* go, float* hid, float* gz, int64_t E, int32_t S, int32_t Sout,
                                     gml_stream_t stream) {
    if (E < 0 || S <= 0 || Sout <= 0) return GML_E_BADARG;
    const int m = S > Sout ? S : Sout;
    if (m <= 16 || m > 48) return GML_E_UNSUPPORTED;
    if (E == 0) return GML_OK;
    if (!ea || !w1 || !w2 || !w3 || !w4 || !out || !gout || !go || !hid || !gz) return GML_E_BADARG;
    if ((S % 4 == 0 && ((uintptr_t)ea & 15)) || (Sout % 4 == 0 && (((uintptr_t)out | (uintptr_t)gout | (uintptr_t)go) & 15)) ||
        (((uintptr_t)hid | (uintptr_t)gz) & 15))
        return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int sp = (m + 3) & ~3;
    switch (sp) {
#define GML_WIDE_BWD(SPV) case SPV: return launch_wide_bwd<SPV>(ea, w1, w2, w3, w4, out, gout, go, hid, gz, E, S, Sout, st);
        GML_WIDE_BWD(20) GML_WIDE_BWD(24) GML_WIDE_BWD(28) GML_WIDE_BWD(32) GML_WIDE_BWD(36) GML_WIDE_BWD(40) GML_WIDE_BWD(44) GML_WIDE_BWD(48)
    }
    return GML_E_UNSUPPORTED;
}
