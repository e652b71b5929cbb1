// ML3Layer Hadamard branch (libs/spect_conv.py:198-202,209), relu-backward glue, segment pooling,
// row gather/scatter, library identification.
#include "gml_common.h"

extern "C" int gml_version(void) { return 1; }

extern "C" const char* gml_error_string(int code) {
    switch (code) {
        case GML_OK: return "ok";
        case GML_E_BADARG: return "gml: bad argument (null pointer, negative size, misaligned or too-small stride)";
        case GML_E_UNSUPPORTED: return "gml: shape outside the compiled kernel set";
        case GML_E_WORKSPACE: return "gml: workspace too small";
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "gml: unknown error";
}

// ---------------------------------------------------------------------------------------------
// out[r, o] = tanh(x[r] . w11[o] + b11[o]) * tanh(x[r] . w12[o] + b12[o])
// One row per GW-lane group, lane <-> output column (strided when F2 > GW).  x row is read by all
// lanes of the group at the same address (broadcast), weights are L1/L2 resident.
// ---------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void gml_k_node_mix(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ w11, const float* __restrict__ b11,
                                                     const float* __restrict__ w12, const float* __restrict__ b12,
                                                     const float* __restrict__ gout, int64_t ldg,
                                                     float* __restrict__ out, int64_t ldo, int64_t nrows, int Fin,
                                                     int F2, int gw) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = tid / gw;
    const int lo = (int)(tid % gw);
    if (row >= nrows) return;
    const float* xr = x + row * ldx;
    for (int o = lo; o < F2; o += gw) {
        float a = b11 ? b11[o] : 0.f, b = b12 ? b12[o] : 0.f;
        const float* wa = w11 + (int64_t)o * Fin;
        const float* wb = w12 + (int64_t)o * Fin;
        for (int f = 0; f < Fin; ++f) {
            const float xv = xr[f];
            a = fmaf(xv, wa[f], a);
            b = fmaf(xv, wb[f], b);
        }
        const float ta = tanhf(a), tb = tanhf(b);
        if constexpr (!BWD) {
            out[row * ldo + o] = ta * tb;
        } else {
            const float g = gout[row * ldg + o];
            out[row * ldo + o] = g * tb * (1.f - ta * ta);
            out[row * ldo + F2 + o] = g * ta * (1.f - tb * tb);
        }
    }
}

static int node_mix_gw(int F2) { return F2 <= 2 ? 2 : (F2 <= 4 ? 4 : (F2 <= 8 ? 8 : (F2 <= 16 ? 16 : (F2 <= 32 ? 32 : 64)))); }

extern "C" int gml_node_mix_fwd(const float* x, int64_t ldx, const float* w11, const float* b11, const float* w12,
                                const float* b12, float* out, int64_t ldo, int64_t num_rows, int32_t Fin,
                                int32_t F2, gml_stream_t stream) {
    if (num_rows < 0 || Fin <= 0 || F2 <= 0 || ldx < Fin || ldo < F2) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!x || !w11 || !w12 || !out) return GML_E_BADARG;
    const int gw = node_mix_gw(F2);
    const int64_t threads = num_rows * gw;
    hipLaunchKernelGGL((gml_k_node_mix<false>), dim3((unsigned)gml_cdiv(threads, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, w11, b11, w12, b12, (const float*)nullptr, (int64_t)0, out, ldo,
                       num_rows, Fin, F2, gw);
    return gml_launch_status();
}

extern "C" int gml_node_mix_bwd(const float* x, int64_t ldx, const float* w11, const float* b11, const float* w12,
                                const float* b12, const float* gout, int64_t ldg, float* gz, int64_t num_rows,
                                int32_t Fin, int32_t F2, gml_stream_t stream) {
    if (num_rows < 0 || Fin <= 0 || F2 <= 0 || ldx < Fin || ldg < F2) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!x || !w11 || !w12 || !gout || !gz) return GML_E_BADARG;
    const int gw = node_mix_gw(F2);
    const int64_t threads = num_rows * gw;
    hipLaunchKernelGGL((gml_k_node_mix<true>), dim3((unsigned)gml_cdiv(threads, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, w11, b11, w12, b12, gout, ldg, gz, (int64_t)(2 * F2),
                       num_rows, Fin, F2, gw);
    return gml_launch_status();
}

// ---------------------------------------------------------------------------------------------
__global__ void gml_k_relu_bwd(const float* __restrict__ gy, int64_t ldgy, const float* __restrict__ y, int64_t ldy,
                               float* __restrict__ g, int64_t ldg, int64_t nrows, int F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * F) return;
    const int64_t r = i / F;
    const int c = (int)(i % F);
    g[r * ldg + c] = (y[r * ldy + c] > 0.f) ? gy[r * ldgy + c] : 0.f;
}

extern "C" int gml_relu_bwd(const float* gy, int64_t ldgy, const float* y, int64_t ldy, float* g, int64_t ldg,
                            int64_t num_rows, int32_t F, gml_stream_t stream) {
    if (num_rows < 0 || F <= 0 || ldgy < F || ldy < F || ldg < F) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!gy || !y || !g) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_relu_bwd, dim3((unsigned)gml_cdiv(num_rows * F, 256)), dim3(256), 0, (hipStream_t)stream,
                       gy, ldgy, y, ldy, g, ldg, num_rows, F);
    return gml_launch_status();
}

// ---------------------------------------------------------------------------------------------
// out[g, :] = sum (or mean) of x rows [ptr[g], ptr[g+1]) -- rows summed in ascending order like the
// CPU index_add_ of global_add_pool.  One lane per (segment, column).
// ---------------------------------------------------------------------------------------------
__global__ void gml_k_segment_sum(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
                                  float* __restrict__ out, int64_t ldo, int64_t nseg, int F, int mean) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseg * F) return;
    const int64_t g = i / F;
    const int c = (int)(i % F);
    const int r0 = ptr[g], r1 = ptr[g + 1];
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += x[(int64_t)r * ldx + c];
    if (mean) a = a / (float)max(r1 - r0, 1);
    out[g * ldo + c] = a;
}

extern "C" int gml_segment_sum(const float* x, int64_t ldx, const int32_t* ptr, float* out, int64_t ldo,
                               int64_t num_segments, int32_t F, int32_t mean, gml_stream_t stream) {
    if (num_segments < 0 || F <= 0 || ldx < F || ldo < F) return GML_E_BADARG;
    if (num_segments == 0) return GML_OK;
    if (!x || !ptr || !out) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_segment_sum, dim3((unsigned)gml_cdiv(num_segments * F, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, ptr, out, ldo, num_segments, F, mean);
    return gml_launch_status();
}

// ---------------------------------------------------------------------------------------------
template <bool SCATTER>
__global__ void gml_k_perm_rows(const float* __restrict__ in, const int32_t* __restrict__ perm,
                                float* __restrict__ out, int64_t rows, int width) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * width) return;
    const int64_t r = i / width;
    const int c = (int)(i % width);
    const int64_t pr = perm[r];
    if constexpr (SCATTER) out[pr * width + c] = in[i];
    else out[i] = in[pr * width + c];
}

extern "C" int gml_gather_rows(const float* in, const int32_t* perm, float* out, int64_t rows, int32_t width,
                               gml_stream_t stream) {
    if (rows < 0 || width <= 0) return GML_E_BADARG;
    if (rows == 0) return GML_OK;
    if (!in || !perm || !out) return GML_E_BADARG;
    hipLaunchKernelGGL((gml_k_perm_rows<false>), dim3((unsigned)gml_cdiv(rows * width, 256)), dim3(256), 0,
                       (hipStream_t)stream, in, perm, out, rows, width);
    return gml_launch_status();
}

extern "C" int gml_scatter_rows(const float* in, const int32_t* perm, float* out, int64_t rows, int32_t width,
                                gml_stream_t stream) {
    if (rows < 0 || width <= 0) return GML_E_BADARG;
    if (rows == 0) return GML_OK;
    if (!in || !perm || !out) return GML_E_BADARG;
    hipLaunchKernelGGL((gml_k_perm_rows<true>), dim3((unsigned)gml_cdiv(rows * width, 256)), dim3(256), 0,
                       (hipStream_t)stream, in, perm, out, rows, width);
    return gml_launch_status();
}
