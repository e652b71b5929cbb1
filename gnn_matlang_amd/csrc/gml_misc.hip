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
// ML3Layer Hadamard branch:  out[r, o] = tanh(x[r] . w11[o] + b11[o]) * tanh(x[r] . w12[o] + b12[o])
// Tiles of 64 rows: the x tile and both weight matrices sit in LDS (coalesced loads), one lane per
// (row, o) pair.  Backward recomputes the tanh's, keeps gz = [dL/dz11 | dL/dz12] of the tile in LDS and
// from it forms, in the same launch, dx += gz [w11; w12], and the weight / bias gradients as
// per-lane register accumulators over all tiles of the (persistent) workgroup -> one partial per
// workgroup, folded in fixed order by gml_k_reduce_rows_misc (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
#define NM_TILE 64
#define NM_NA 16

template <bool BWD>
__global__ __launch_bounds__(256) void gml_k_node_mix(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ w11, const float* __restrict__ b11,
                                                     const float* __restrict__ w12, const float* __restrict__ b12,
                                                     const float* __restrict__ gout, int64_t ldg,
                                                     float* __restrict__ out, int64_t ldo,
                                                     float* __restrict__ dx, int64_t lddx, float* __restrict__ partial,
                                                     int64_t nrows, int Fin, int F2, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LDX = Fin | 1;                                  // odd: conflict-free row-strided reads
    const int C2 = 2 * F2, LDZ = C2 | 1;
    float* xs = lds;                                          // [64][LDX]
    float* wc = xs + NM_TILE * LDX;                           // [2*F2][LDX]  rows: w11 then w12
    float* bc = wc + C2 * LDX;                                // [2*F2]
    float* gz = bc + C2;                                      // [64][LDZ]   (backward only)
    const int tid = threadIdx.x;
    for (int i = tid; i < C2 * Fin; i += 256) {
        const int c = i / Fin, f = i % Fin;
        wc[c * LDX + f] = (c < F2) ? w11[c * Fin + f] : w12[(c - F2) * Fin + f];
    }
    for (int i = tid; i < C2; i += 256) bc[i] = (i < F2) ? (b11 ? b11[i] : 0.f) : (b12 ? b12[i - F2] : 0.f);

    const int npair = C2 * Fin + C2;                          // weight entries + bias entries
    float acc[NM_NA];
#pragma unroll
    for (int a = 0; a < NM_NA; ++a) acc[a] = 0.f;

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t r0 = (int64_t)t * NM_TILE;
        const int nr = (int)min((int64_t)NM_TILE, nrows - r0);
        __syncthreads();
        for (int i = tid; i < NM_TILE * Fin; i += 256) {
            const int rr = i / Fin, f = i % Fin;
            xs[rr * LDX + f] = (rr < nr) ? x[(r0 + rr) * ldx + f] : 0.f;
        }
        __syncthreads();
        for (int pidx = tid; pidx < NM_TILE * F2; pidx += 256) {
            const int rr = pidx / F2, o = pidx % F2;
            float a = bc[o], b = bc[F2 + o];
            const float* xr = xs + rr * LDX;
            const float* wa = wc + o * LDX;
            const float* wb = wc + (F2 + o) * LDX;
            for (int f = 0; f < Fin; ++f) {
                a = fmaf(xr[f], wa[f], a);
                b = fmaf(xr[f], wb[f], b);
            }
            const float ta = tanhf(a), tb = tanhf(b);
            if constexpr (!BWD) {
                if (rr < nr) out[(r0 + rr) * ldo + o] = ta * tb;
            } else {
                const float g = (rr < nr) ? gout[(r0 + rr) * ldg + o] : 0.f;
                gz[rr * LDZ + o] = g * tb * (1.f - ta * ta);
                gz[rr * LDZ + F2 + o] = g * ta * (1.f - tb * tb);
            }
        }
        if constexpr (BWD) {
            __syncthreads();
            if (dx) {
                for (int q = tid; q < nr * Fin; q += 256) {
                    const int rr = q / Fin, f = q % Fin;
                    float a = 0.f;
                    for (int c = 0; c < C2; ++c) a = fmaf(gz[rr * LDZ + c], wc[c * LDX + f], a);
                    dx[(r0 + rr) * lddx + f] += a;
                }
            }
#pragma unroll
            for (int a = 0; a < NM_NA; ++a) {
                const int pidx = tid + 256 * a;
                if (pidx < npair) {
                    float s = acc[a];
                    if (pidx < C2 * Fin) {
                        const int c = pidx / Fin, f = pidx % Fin;
                        for (int rr = 0; rr < NM_TILE; ++rr) s = fmaf(gz[rr * LDZ + c], xs[rr * LDX + f], s);
                    } else {
                        const int c = pidx - C2 * Fin;
                        for (int rr = 0; rr < NM_TILE; ++rr) s += gz[rr * LDZ + c];
                    }
                    acc[a] = s;
                }
            }
        }
    }
    if constexpr (BWD) {
#pragma unroll
        for (int a = 0; a < NM_NA; ++a) {
            const int pidx = tid + 256 * a;
            if (pidx < npair) partial[(int64_t)blockIdx.x * npair + pidx] = acc[a];
        }
    }
}

// fold [nparts][n] partials in fixed order and split into dw11 | dw12 | db11 | db12
__global__ void gml_k_node_mix_fold(const float* __restrict__ partial, int64_t nparts, int Fin, int F2,
                                    float* __restrict__ dw11, float* __restrict__ db11, float* __restrict__ dw12,
                                    float* __restrict__ db12) {
    const int n = 2 * F2 * Fin + 2 * F2;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float a = 0.f;
    for (int64_t w = 0; w < nparts; ++w) a += partial[w * n + j];
    if (j < F2 * Fin) dw11[j] = a;
    else if (j < 2 * F2 * Fin) dw12[j - F2 * Fin] = a;
    else if (j < 2 * F2 * Fin + F2) { if (db11) db11[j - 2 * F2 * Fin] = a; }
    else { if (db12) db12[j - 2 * F2 * Fin - F2] = a; }
}

static size_t node_mix_lds(int Fin, int F2, bool bwd) {
    const int LDX = Fin | 1, C2 = 2 * F2, LDZ = C2 | 1;
    return sizeof(float) * (size_t)(NM_TILE * LDX + C2 * LDX + C2 + (bwd ? NM_TILE * LDZ : 0));
}

static int node_mix_grid(int64_t num_rows) {
    const int64_t nt = gml_cdiv(num_rows, NM_TILE);
    return (int)(nt < GML_NUM_CU * 4 ? nt : GML_NUM_CU * 4);
}

extern "C" int gml_node_mix_fwd(const float* x, int64_t ldx, const float* w11, const float* b11, const float* w12,
                                const float* b12, float* out, int64_t ldo, int64_t num_rows, int32_t Fin,
                                int32_t F2, gml_stream_t stream) {
    if (num_rows < 0 || Fin <= 0 || F2 <= 0 || ldx < Fin || ldo < F2) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!x || !w11 || !w12 || !out) return GML_E_BADARG;
    const size_t lds = node_mix_lds(Fin, F2, false);
    if (lds > 64 * 1024) return GML_E_UNSUPPORTED;
    hipLaunchKernelGGL((gml_k_node_mix<false>), dim3(node_mix_grid(num_rows)), dim3(256), lds, (hipStream_t)stream, x,
                       ldx, w11, b11, w12, b12, (const float*)nullptr, (int64_t)0, out, ldo, (float*)nullptr,
                       (int64_t)0, (float*)nullptr, num_rows, Fin, F2, (int)gml_cdiv(num_rows, NM_TILE));
    return gml_launch_status();
}

extern "C" size_t gml_node_mix_bwd_workspace_bytes(int64_t num_rows, int32_t Fin, int32_t F2) {
    if (num_rows <= 0 || Fin <= 0 || F2 <= 0) return 0;
    const int npair = 2 * F2 * Fin + 2 * F2;
    if (npair > 256 * NM_NA || node_mix_lds(Fin, F2, true) > 64 * 1024) return 0;   /* 0 = shape not supported */
    return sizeof(float) * (size_t)node_mix_grid(num_rows) * npair;
}

extern "C" int gml_node_mix_bwd(const float* x, int64_t ldx, const float* w11, const float* b11, const float* w12,
                                const float* b12, const float* gout, int64_t ldg, float* dx, int64_t lddx,
                                float* dw11, float* db11, float* dw12, float* db12, int64_t num_rows, int32_t Fin,
                                int32_t F2, void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (num_rows < 0 || Fin <= 0 || F2 <= 0 || ldx < Fin || ldg < F2 || (dx && lddx < Fin)) return GML_E_BADARG;
    if (!w11 || !w12 || !dw11 || !dw12) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (num_rows == 0) {
        hipMemsetAsync(dw11, 0, sizeof(float) * F2 * Fin, st);
        hipMemsetAsync(dw12, 0, sizeof(float) * F2 * Fin, st);
        if (db11) hipMemsetAsync(db11, 0, sizeof(float) * F2, st);
        if (db12) hipMemsetAsync(db12, 0, sizeof(float) * F2, st);
        return gml_launch_status();
    }
    if (!x || !gout) return GML_E_BADARG;
    const size_t need = gml_node_mix_bwd_workspace_bytes(num_rows, Fin, F2);
    if (need == 0) return GML_E_UNSUPPORTED;
    if (!ws || ws_bytes < need) return GML_E_WORKSPACE;
    const int grid = node_mix_grid(num_rows);
    hipLaunchKernelGGL((gml_k_node_mix<true>), dim3(grid), dim3(256), node_mix_lds(Fin, F2, true), st, x, ldx, w11,
                       b11, w12, b12, gout, ldg, (float*)nullptr, (int64_t)0, dx, lddx, (float*)ws, num_rows, Fin,
                       F2, (int)gml_cdiv(num_rows, NM_TILE));
    int rc = gml_launch_status();
    if (rc != GML_OK) return rc;
    const int n = 2 * F2 * Fin + 2 * F2;
    hipLaunchKernelGGL(gml_k_node_mix_fold, dim3((unsigned)gml_cdiv(n, 256)), dim3(256), 0, st, (const float*)ws,
                       (int64_t)grid, Fin, F2, dw11, db11, dw12, db12);
    return gml_launch_status();
}

// ---------------------------------------------------------------------------------------------
// columns [F, ldg) of g are zero-filled so that consumers may read whole aligned float4 groups
__global__ void gml_k_relu_bwd(const float* __restrict__ gy, int64_t ldgy, const float* __restrict__ y, int64_t ldy,
                               float* __restrict__ g, int64_t ldg, int64_t nrows, int F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * ldg) return;
    const int64_t r = i / ldg;
    const int c = (int)(i % ldg);
    g[i] = (c < F && y[r * ldy + c] > 0.f) ? gy[r * ldgy + c] : 0.f;
}

extern "C" int gml_relu_bwd(const float* gy, int64_t ldgy, const float* y, int64_t ldy, float* g, int64_t ldg,
                            int64_t num_rows, int32_t F, gml_stream_t stream) {
    if (num_rows < 0 || F <= 0 || ldgy < F || ldy < F || ldg < F) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!gy || !y || !g) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_relu_bwd, dim3((unsigned)gml_cdiv(num_rows * ldg, 256)), dim3(256), 0, (hipStream_t)stream,
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
