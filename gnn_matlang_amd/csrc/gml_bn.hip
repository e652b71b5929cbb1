// Batch normalisation over the rows of x [N, C] in training mode (torch.nn.BatchNorm1d between the layers of mutag.py:272-288,
// mnist75.py's readout): batch statistics, normalise + affine, and the backward, as four bandwidth-shaped launches.
// Round 4: on [602 k, 48] torch's four kernels took 98 + 74 + 129 + 106 us per layer (1.2 - 2.0 TB/s) -- a third of the mutag step.
//   stats      column sums of x and x^2 over row chunks (float4 per lane, 16 quads x 16 rows per sweep) -> per-block partials,
//              combined in double by one small block: mean, biased variance, 1 / sqrt(var + eps)
//   apply      y = (x - mean) rstd w + b
//   bwd_sums   column sums of dy and dy xhat (xhat recomputed from x): = d bias and d weight
//   bwd_apply  dx = (dy - sum_dy / N - xhat sum_dyxhat / N) rstd w
// C <= 64, C % 4 == 0, float4-addressable rows (the layer outputs of this package); anything else: GML_E_UNSUPPORTED (the caller keeps
// torch's implementation).  The variance is E[x^2] - mean^2 with the partial sums combined in double: fine for activations whose mean
// is of the order of their spread (post-relu layer outputs); torch's Welford pass is the safer choice for |mean| >> std inputs.
#include "gml_common.h"

#define GML_BN_ROWS 2048          /* rows per block of the two reduction kernels */

// mode 0: s0 = sum x, s1 = sum x^2.  mode 1: a = dy, b = x: s0 = sum dy, s1 = sum dy * (x - mean) rstd
template <int MODE>
__global__ __launch_bounds__(256) void gml_k_bn_colsums(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                        int64_t N, int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        float* __restrict__ partial) {
    __shared__ f32x4 red[2][16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c4 = 4 * tx;
    const bool on = c4 < C;
    f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0, mu = s0, rs = s0;
    if (MODE == 1 && on) { mu = *reinterpret_cast<const f32x4*>(mean + c4); rs = *reinterpret_cast<const f32x4*>(rstd + c4); }
    const int64_t r0 = (int64_t)blockIdx.x * GML_BN_ROWS;
    const int64_t r1 = r0 + GML_BN_ROWS < N ? r0 + GML_BN_ROWS : N;
    if (on) {
        for (int64_t r = r0 + ty; r < r1; r += 64) {           // four independent loads in flight per thread
            f32x4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = r + 16 * u < r1 ? r + 16 * u : r1 - 1;
                va[u] = *reinterpret_cast<const f32x4*>(a + rr * lda + c4);
                if (MODE == 1) vb[u] = *reinterpret_cast<const f32x4*>(b + rr * ldb + c4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + 16 * u < r1) {
                    s0 += va[u];
                    if (MODE == 0) s1 += va[u] * va[u];
                    else s1 += va[u] * ((vb[u] - mu) * rs);
                }
        }
    }
    red[0][ty][tx] = s0;
    red[1][ty][tx] = s1;
    __syncthreads();
    if (threadIdx.x < 32) {                                   // 16 quads x {s0, s1}
        const int q = threadIdx.x & 15, w = threadIdx.x >> 4;
        f32x4 t = red[w][0][q];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[w][k][q];
        *reinterpret_cast<f32x4*>(partial + ((int64_t)blockIdx.x * 2 + w) * 64 + 4 * q) = t;
    }
}

// one block of 1024 threads: 16 thread groups share the partials of a column (a single chain of ~300 dependent loads per column
// took 72 us -- more than the reduction kernel it follows), combined in double
template <int MODE>
__global__ __launch_bounds__(1024) void gml_k_bn_finish(const float* __restrict__ partial, int nblk, int64_t N, int C, float eps,
                                                        float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2) {
    __shared__ double red[2][16][64];
    const int c = threadIdx.x & 63, j = threadIdx.x >> 6;
    double s0 = 0.0, s1 = 0.0;
    for (int k = j; k < nblk; k += 16) { s0 += (double)partial[((int64_t)k * 2) * 64 + c]; s1 += (double)partial[((int64_t)k * 2 + 1) * 64 + c]; }
    red[0][j][c] = s0;
    red[1][j][c] = s1;
    __syncthreads();
    if (j != 0 || c >= C) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) { s0 += red[0][k][c]; s1 += red[1][k][c]; }
    if (MODE == 0) {
        const double m = s0 / (double)N;
        double v = s1 / (double)N - m * m;
        if (v < 0.0) v = 0.0;
        o0[c] = (float)m; o1[c] = (float)v; o2[c] = (float)(1.0 / sqrt(v + (double)eps));
    } else {
        o0[c] = (float)s0; o1[c] = (float)s1;
    }
}

__global__ __launch_bounds__(256) void gml_k_bn_apply(const float* __restrict__ x, int64_t ldx, int64_t N, int C, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ y, int64_t ldy) {
    const int cq = C >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * cq) return;
    const int64_t r = i / cq;
    const int c4 = (int)(i - r * cq) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4), rs = *reinterpret_cast<const f32x4*>(rstd + c4);
    f32x4 o = (v - mu) * rs;
    if (w) o = o * *reinterpret_cast<const f32x4*>(w + c4);
    if (bias) o = o + *reinterpret_cast<const f32x4*>(bias + c4);
    *reinterpret_cast<f32x4*>(y + r * ldy + c4) = o;
}

__global__ __launch_bounds__(256) void gml_k_bn_bwd_apply(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                          int64_t N, int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ w, const float* __restrict__ sdy, const float* __restrict__ sdyx,
                                                          float* __restrict__ dx, int64_t lddx) {
    const int cq = C >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * cq) return;
    const int64_t r = i / cq;
    const int c4 = (int)(i - r * cq) * 4;
    const float invn = 1.f / (float)N;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * lddy + c4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4), rs = *reinterpret_cast<const f32x4*>(rstd + c4);
    const f32x4 a = *reinterpret_cast<const f32x4*>(sdy + c4) * invn, b = *reinterpret_cast<const f32x4*>(sdyx + c4) * invn;
    const f32x4 xh = (v - mu) * rs;
    f32x4 o = (g - a - xh * b) * rs;
    if (w) o = o * *reinterpret_cast<const f32x4*>(w + c4);
    *reinterpret_cast<f32x4*>(dx + r * lddx + c4) = o;
}

static bool bn_shape_ok(const void* p, int64_t ld, int32_t C) { return C > 0 && C <= 64 && C % 4 == 0 && ld % 4 == 0 && ld >= C && (((uintptr_t)p) & 15) == 0; }

extern "C" size_t gml_bn_workspace_bytes(int64_t num_rows) { return num_rows <= 0 ? 0 : (size_t)gml_cdiv(num_rows, GML_BN_ROWS) * 2 * 64 * sizeof(float); }

extern "C" int gml_bn_stats(const float* x, int64_t ldx, int64_t num_rows, int32_t C, float eps, float* mean, float* var, float* rstd,
                            void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (num_rows <= 0 || !x || !mean || !var || !rstd) return GML_E_BADARG;
    if (!bn_shape_ok(x, ldx, C)) return GML_E_UNSUPPORTED;
    if (!ws || ws_bytes < gml_bn_workspace_bytes(num_rows)) return GML_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)gml_cdiv(num_rows, GML_BN_ROWS);
    hipLaunchKernelGGL(gml_k_bn_colsums<0>, dim3(nblk), dim3(256), 0, st, x, ldx, (const float*)nullptr, (int64_t)0, num_rows, (int)C,
                       (const float*)nullptr, (const float*)nullptr, (float*)ws);
    hipLaunchKernelGGL(gml_k_bn_finish<0>, dim3(1), dim3(1024), 0, st, (const float*)ws, nblk, num_rows, (int)C, eps, mean, var, rstd);
    return gml_launch_status();
}

extern "C" int gml_bn_apply(const float* x, int64_t ldx, int64_t num_rows, int32_t C, const float* mean, const float* rstd, const float* weight,
                            const float* bias, float* y, int64_t ldy, gml_stream_t stream) {
    if (num_rows <= 0 || !x || !mean || !rstd || !y) return GML_E_BADARG;
    if (!bn_shape_ok(x, ldx, C) || !bn_shape_ok(y, ldy, C)) return GML_E_UNSUPPORTED;
    hipLaunchKernelGGL(gml_k_bn_apply, dim3((unsigned)gml_cdiv(num_rows * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, num_rows,
                       (int)C, mean, rstd, weight, bias, y, ldy);
    return gml_launch_status();
}

extern "C" int gml_bn_bwd_sums(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t num_rows, int32_t C, const float* mean,
                               const float* rstd, float* sum_dy, float* sum_dyxhat, void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (num_rows <= 0 || !dy || !x || !mean || !rstd || !sum_dy || !sum_dyxhat) return GML_E_BADARG;
    if (!bn_shape_ok(x, ldx, C) || !bn_shape_ok(dy, lddy, C)) return GML_E_UNSUPPORTED;
    if (!ws || ws_bytes < gml_bn_workspace_bytes(num_rows)) return GML_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)gml_cdiv(num_rows, GML_BN_ROWS);
    hipLaunchKernelGGL(gml_k_bn_colsums<1>, dim3(nblk), dim3(256), 0, st, dy, lddy, x, ldx, num_rows, (int)C, mean, rstd, (float*)ws);
    hipLaunchKernelGGL(gml_k_bn_finish<1>, dim3(1), dim3(1024), 0, st, (const float*)ws, nblk, num_rows, (int)C, 0.f, sum_dy, sum_dyxhat,
                       (float*)nullptr);
    return gml_launch_status();
}

extern "C" int gml_bn_bwd_apply(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t num_rows, int32_t C, const float* mean,
                                const float* rstd, const float* weight, const float* sum_dy, const float* sum_dyxhat, float* dx, int64_t lddx,
                                gml_stream_t stream) {
    if (num_rows <= 0 || !dy || !x || !mean || !rstd || !sum_dy || !sum_dyxhat || !dx) return GML_E_BADARG;
    if (!bn_shape_ok(x, ldx, C) || !bn_shape_ok(dy, lddy, C) || !bn_shape_ok(dx, lddx, C)) return GML_E_UNSUPPORTED;
    hipLaunchKernelGGL(gml_k_bn_bwd_apply, dim3((unsigned)gml_cdiv(num_rows * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx,
                       num_rows, (int)C, mean, rstd, weight, sum_dy, sum_dyxhat, dx, lddx);
    return gml_launch_status();
}
