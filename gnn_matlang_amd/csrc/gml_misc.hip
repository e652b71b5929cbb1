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
// (/root/reference/libs/spect_conv.py:198-202,209).
// One ROW per lane, 256 rows per workgroup step: the x tile is staged through LDS with coalesced loads,
// each lane keeps its row in registers, the weights are LDS broadcast reads.  Backward recomputes the
// tanh's and, in the same launch, adds dx += dz [w11; w12] (row in registers) and contracts the
// weight / bias gradients dz^T [x | 1] over the rows on the matrix cores (v_mfma_f32_16x16x4_f32,
// A = dz tile, B = x tile, both already in LDS), accumulating per wave across all its tiles; one
// partial per wave, folded in fixed order (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
#define NM_ROWS 256
#define NM_MAXCB 3      /* 2*F2 <= 48 */

template <int FINP, bool BWD>
__global__ __launch_bounds__(256) void gml_k_node_mix(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ w11, const float* __restrict__ b11,
                                                     const float* __restrict__ w12, const float* __restrict__ b12,
                                                     const float* __restrict__ gout, int64_t ldg,
                                                     float* __restrict__ out, int64_t ldo,
                                                     float* __restrict__ dx, int64_t lddx, float* __restrict__ partial,
                                                     int64_t nrows, int Fin, int F2, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LDX = FINP + 1;                              // odd: row-per-lane reads are conflict free
    constexpr int NFB = FINP / 16;
    const int C2 = 2 * F2, C2P = (C2 + 15) / 16 * 16, LDZ = C2P + 1, ncb = C2P / 16;
    float* xs = lds;                                           // [256][LDX]
    float* wc = xs + NM_ROWS * LDX;                            // [C2][FINP]  rows: w11 then w12, zero padded
    float* bc = wc + C2 * FINP;                                // [C2]
    float* gz = bc + ((C2 + 3) / 4 * 4);                       // [256][LDZ]  (backward only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    for (int i = tid; i < C2 * FINP; i += 256) {
        const int c = i / FINP, f = i % FINP;
        wc[i] = (f < Fin) ? ((c < F2) ? w11[c * Fin + f] : w12[(c - F2) * Fin + f]) : 0.f;
    }
    for (int i = tid; i < C2; i += 256) bc[i] = (i < F2) ? (b11 ? b11[i] : 0.f) : (b12 ? b12[i - F2] : 0.f);

    f32x4 acc[NM_MAXCB][NFB + 1];
#pragma unroll
    for (int a = 0; a < NM_MAXCB; ++a)
#pragma unroll
        for (int b = 0; b <= NFB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t r0 = (int64_t)t * NM_ROWS;
        const int nr = (int)min((int64_t)NM_ROWS, nrows - r0);
        __syncthreads();
        if ((ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0) {    // float4 rows: lane <-> 4 features, clamped unconditional loads
            constexpr int FV = FINP / 4, RPS = 256 / FV, NIT = (NM_ROWS + RPS - 1) / RPS;
            const int f = (tid % FV) * 4, rbase = tid / FV, fc = min(f, (Fin - 1) / 4 * 4);
            f32x4 v[NIT];
#pragma unroll
            for (int j = 0; j < NIT; ++j)
                v[j] = *reinterpret_cast<const f32x4*>(x + (r0 + min(rbase + j * RPS, nr - 1)) * ldx + fc);
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int rr = rbase + j * RPS;
                if (rr < NM_ROWS) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) xs[rr * LDX + f + k] = (rr < nr && f + k < Fin) ? v[j][k] : 0.f;
                }
            }
        } else {                                               // coalesced tile load: lane <-> feature, 256/FINP rows
            constexpr int RPS = 256 / FINP, NIT = (NM_ROWS + RPS - 1) / RPS;   // per sweep; ALL loads in flight
            const int f = tid % FINP, rbase = tid / FINP, fc = min(f, Fin - 1);
            float v[NIT];
#pragma unroll
            for (int j = 0; j < NIT; ++j) v[j] = x[(r0 + min(rbase + j * RPS, nr - 1)) * ldx + fc];
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int rr = rbase + j * RPS;
                if (rr < NM_ROWS) xs[rr * LDX + f] = (rr < nr && f < Fin) ? v[j] : 0.f;
            }
        }
        __syncthreads();
        const bool rv = tid < nr;
        float xr[FINP];
#pragma unroll
        for (int f = 0; f < FINP; ++f) xr[f] = xs[tid * LDX + f];
        float dxr[BWD ? FINP : 1];
        if constexpr (BWD) {
#pragma unroll
            for (int f = 0; f < FINP; ++f) dxr[f] = 0.f;
            for (int c = C2; c < C2P; ++c) gz[tid * LDZ + c] = 0.f;               // zero the padding columns
        }
        for (int o = 0; o < F2; ++o) {
            float a = bc[o], b = bc[F2 + o];
            const float* wa = wc + o * FINP;
            const float* wb = wc + (F2 + o) * FINP;
#pragma unroll
            for (int f4 = 0; f4 < FINP / 4; ++f4) {
                const f32x4 va = *reinterpret_cast<const f32x4*>(wa + 4 * f4);
                const f32x4 vb = *reinterpret_cast<const f32x4*>(wb + 4 * f4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a = fmaf(xr[4 * f4 + i], va[i], a);
                    b = fmaf(xr[4 * f4 + i], vb[i], b);
                }
            }
            float ta, tb, da, db;
            gml_tanh_d(a, ta, da);
            gml_tanh_d(b, tb, db);
            if constexpr (!BWD) {
                if (rv) out[(r0 + tid) * ldo + o] = ta * tb;
            } else {
                const float g = rv ? gout[(r0 + tid) * ldg + o] : 0.f;
                const float g1 = g * tb * da, g2 = g * ta * db;
                gz[tid * LDZ + o] = g1;
                gz[tid * LDZ + F2 + o] = g2;
#pragma unroll
                for (int f4 = 0; f4 < FINP / 4; ++f4) {
                    const f32x4 va = *reinterpret_cast<const f32x4*>(wa + 4 * f4);
                    const f32x4 vb = *reinterpret_cast<const f32x4*>(wb + 4 * f4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dxr[4 * f4 + i] = fmaf(g1, va[i], fmaf(g2, vb[i], dxr[4 * f4 + i]));
                }
            }
        }
        if constexpr (BWD) {
            if (dx && rv) {                                    // read-modify-write of the lane's own row: all loads first
                float* dr = dx + (r0 + tid) * lddx;
                float old[FINP];
#pragma unroll
                for (int f = 0; f < FINP; ++f) old[f] = (f < Fin) ? dr[f] : 0.f;
#pragma unroll
                for (int f = 0; f < FINP; ++f)
                    if (f < Fin) dr[f] = old[f] + dxr[f];
            }
            // weight / bias gradients of this wave's 64 rows: D[c][f] += sum_rows dz[row][c] * [x | 1][row][f]
            const int rb = wave * 64;
#pragma unroll
            for (int t16 = 0; t16 < 16; ++t16) {
                const int rr = rb + 4 * t16 + kq;
#pragma unroll
                for (int cb = 0; cb < NM_MAXCB; ++cb) {
                    if (cb < ncb) {
                        const float a = gz[rr * LDZ + cb * 16 + r16];
#pragma unroll
                        for (int fb = 0; fb < NFB; ++fb)
                            acc[cb][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xs[rr * LDX + fb * 16 + r16], acc[cb][fb], 0, 0, 0);
                        acc[cb][NFB] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, (r16 == 0) ? 1.f : 0.f, acc[cb][NFB], 0, 0, 0);
                    }
                }
            }
        }
    }
    if constexpr (BWD) {
        // D layout: lane (col j = r16, rows i = 4*kq + reg): i = c index, j = f index
        const int npair = C2 * Fin + C2;
        float* P = partial + ((int64_t)blockIdx.x * 4 + wave) * npair;
#pragma unroll
        for (int cb = 0; cb < NM_MAXCB; ++cb) {
            if (cb < ncb) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int c = cb * 16 + 4 * kq + reg;
                    if (c < C2) {
#pragma unroll
                        for (int fb = 0; fb < NFB; ++fb) {
                            const int f = fb * 16 + r16;
                            if (f < Fin) P[c * Fin + f] = acc[cb][fb][reg];
                        }
                        if (r16 == 0) P[C2 * Fin + c] = acc[cb][NFB][reg];
                    }
                }
            }
        }
    }
}

// fold [nparts][n] partials in fixed order and split into dw11 | dw12 | db11 | db12
__global__ __launch_bounds__(256) void gml_k_node_mix_fold(const float* __restrict__ partial, int64_t nparts, int Fin,
                                                          int F2, float* __restrict__ dw11, float* __restrict__ db11,
                                                          float* __restrict__ dw12, float* __restrict__ db12) {
    __shared__ float red[16][17];
    const int n = 2 * F2 * Fin + 2 * F2;
    const int jl = threadIdx.x & 15, wl = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + jl;
    float a = 0.f;
    if (j < n) a = gml_fold_column(partial, nparts, n, j, wl);
    red[wl][jl] = a;
    __syncthreads();
    if (wl != 0 || j >= n) return;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][jl];
    if (j < F2 * Fin) dw11[j] = t;
    else if (j < 2 * F2 * Fin) dw12[j - F2 * Fin] = t;
    else if (j < 2 * F2 * Fin + F2) { if (db11) db11[j - 2 * F2 * Fin] = t; }
    else { if (db12) db12[j - 2 * F2 * Fin - F2] = t; }
}

static int node_mix_finp(int Fin) { return Fin <= 16 ? 16 : (Fin <= 32 ? 32 : (Fin <= 48 ? 48 : (Fin <= 64 ? 64 : 0))); }

static size_t node_mix_lds(int Fin, int F2, bool bwd) {
    const int FINP = node_mix_finp(Fin), C2 = 2 * F2, C2P = (C2 + 15) / 16 * 16;
    return sizeof(float) * (size_t)(NM_ROWS * (FINP + 1) + C2 * FINP + (C2 + 3) / 4 * 4 + (bwd ? NM_ROWS * (C2P + 1) : 0));
}

static int node_mix_grid(int64_t num_rows) {
    const int64_t nt = gml_cdiv(num_rows, NM_ROWS);
    return (int)(nt < GML_NUM_CU * 2 ? nt : GML_NUM_CU * 2);
}

static bool node_mix_supported(int Fin, int F2) { return node_mix_finp(Fin) != 0 && 2 * F2 <= 16 * NM_MAXCB; }

template <bool BWD>
static int node_mix_launch(int FINP, dim3 grid, size_t lds, hipStream_t st, const float* x, int64_t ldx, const float* w11,
                           const float* b11, const float* w12, const float* b12, const float* gout, int64_t ldg,
                           float* out, int64_t ldo, float* dx, int64_t lddx, float* partial, int64_t nrows, int Fin,
                           int F2, int ntiles) {
#define NM_GO(FP)                                                                                             \
    if (FINP == FP) {                                                                                         \
        GML_ALLOW_BIG_LDS(arc, (&gml_k_node_mix<FP, BWD>), 160 * 1024) \
        if (arc != hipSuccess) return (int)arc;                                                               \
        hipLaunchKernelGGL((gml_k_node_mix<FP, BWD>), grid, dim3(256), lds, st, x, ldx, w11, b11, w12, b12, gout, ldg, \
                           out, ldo, dx, lddx, partial, nrows, Fin, F2, ntiles);                              \
        return gml_launch_status();                                                                           \
    }
    NM_GO(16) NM_GO(32) NM_GO(48) NM_GO(64)
    return GML_E_UNSUPPORTED;
}

extern "C" int gml_node_mix_fwd(const float* x, int64_t ldx, const float* w11, const float* b11, const float* w12,
                                const float* b12, float* out, int64_t ldo, int64_t num_rows, int32_t Fin,
                                int32_t F2, gml_stream_t stream) {
    if (num_rows < 0 || Fin <= 0 || F2 <= 0 || ldx < Fin || ldo < F2) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!x || !w11 || !w12 || !out) return GML_E_BADARG;
    if (!node_mix_supported(Fin, F2)) return GML_E_UNSUPPORTED;
    return node_mix_launch<false>(node_mix_finp(Fin), dim3(node_mix_grid(num_rows)), node_mix_lds(Fin, F2, false),
                                  (hipStream_t)stream, x, ldx, w11, b11, w12, b12, nullptr, 0, out, ldo, nullptr, 0,
                                  nullptr, num_rows, Fin, F2, (int)gml_cdiv(num_rows, NM_ROWS));
}

extern "C" size_t gml_node_mix_bwd_workspace_bytes(int64_t num_rows, int32_t Fin, int32_t F2) {
    if (num_rows <= 0 || Fin <= 0 || F2 <= 0 || !node_mix_supported(Fin, F2)) return 0;   /* 0 = not supported */
    return sizeof(float) * (size_t)node_mix_grid(num_rows) * 4 * (2 * F2 * Fin + 2 * F2);
}

extern "C" int gml_node_mix_bwd(const float* x, int64_t ldx, const float* w11, const float* b11, const float* w12,
                                const float* b12, const float* gout, int64_t ldg, float* dx, int64_t lddx,
                                float* dw11, float* db11, float* dw12, float* db12, int64_t num_rows, int32_t Fin,
                                int32_t F2, void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (num_rows < 0 || Fin <= 0 || F2 <= 0 || ldx < Fin || ldg < F2 || (dx && lddx < Fin)) return GML_E_BADARG;
    if (!w11 || !w12 || !dw11 || !dw12) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (num_rows == 0) {
        gml_zero_async(dw11, sizeof(float) * F2 * Fin, st);
        gml_zero_async(dw12, sizeof(float) * F2 * Fin, st);
        if (db11) gml_zero_async(db11, sizeof(float) * F2, st);
        if (db12) gml_zero_async(db12, sizeof(float) * F2, st);
        return gml_launch_status();
    }
    if (!x || !gout) return GML_E_BADARG;
    const size_t need = gml_node_mix_bwd_workspace_bytes(num_rows, Fin, F2);
    if (need == 0) return GML_E_UNSUPPORTED;
    if (!ws || ws_bytes < need) return GML_E_WORKSPACE;
    const int grid = node_mix_grid(num_rows);
    int rc = node_mix_launch<true>(node_mix_finp(Fin), dim3(grid), node_mix_lds(Fin, F2, true), st, x, ldx, w11, b11,
                                   w12, b12, gout, ldg, nullptr, 0, dx, lddx, (float*)ws, num_rows, Fin, F2,
                                   (int)gml_cdiv(num_rows, NM_ROWS));
    if (rc != GML_OK) return rc;
    const int n = 2 * F2 * Fin + 2 * F2;
    hipLaunchKernelGGL(gml_k_node_mix_fold, dim3((unsigned)gml_cdiv(n, 16)), dim3(256), 0, st, (const float*)ws,
                       (int64_t)grid * 4, Fin, F2, dw11, db11, dw12, db12);
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
    if ((mean & GML_POOL_SKIP_LAST) && g == nseg - 1) { out[g * ldo + c] = 0.f; return; }   // the padding graph of a static batch
    const int r0 = ptr[g], r1 = ptr[g + 1];
    float a = 0.f;
    // same ascending order; 32 clamped loads in flight per trip (8 before: a 1,700-row segment -- the padding graph of a static
    // batch, or a proteins-size graph -- was ~200 dependent trips of one lane; a 23-row molecule is one trip instead of three)
    constexpr int U = 32;
    for (int r = r0; r < r1; r += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[(int64_t)min(r + u, r1 - 1) * ldx + c];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + u < r1) a += v[u];
    }
    if (mean & 1) a = a / (float)max(r1 - r0, 1);
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

// gml_k_segment_sum for 32-column rows that also leaves bit c = (x[row][c] > 0) of every row in mask[row]: the 32 lanes that sum a
// segment hold a whole row per step, one ballot is the row's relu pattern.  The layer in front of the pool is an ML3Layer whose
// output is [relu(conv) | Hadamard columns] (Zinc12k.py:338-343): its backward then needs 4 bytes per row instead of the saved
// output (gml_segment_bcast_mask).  Same sums, in the same order, as gml_k_segment_sum.
__global__ __launch_bounds__(256) void gml_k_segment_sum_mask32(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
                                                                float* __restrict__ out, int64_t ldo, uint32_t* __restrict__ mask,
                                                                int64_t nseg, int mean) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g = i >> 5;
    const int c = (int)(i & 31);
    const bool live = g < nseg;
    const bool pad = live && (mean & GML_POOL_SKIP_LAST) && g == nseg - 1;     // the padding graph of a static batch: zeros, rows not read
    const int r0 = (live && !pad) ? ptr[g] : 0, r1 = (live && !pad) ? ptr[g + 1] : 0;
    const bool upper = (threadIdx.x & 32) != 0;              // which half of the wave's ballot is this segment's
    float a = 0.f;
    constexpr int U = 32;
    for (int r = r0; r < r1; r += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[(int64_t)min(r + u, r1 - 1) * ldx + c];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + u < r1) {
                a += v[u];
                const unsigned long long b = __ballot(v[u] > 0.f);           // (only the lanes inside the branch vote)
                const uint32_t m = upper ? (uint32_t)(b >> 32) : (uint32_t)b;
                if (c == 0) mask[r + u] = m;
            }
    }
    if (!live) return;
    if (mean & 1) a = a / (float)max(r1 - r0, 1);
    out[g * ldo + c] = a;
}

extern "C" int gml_segment_sum_mask(const float* x, int64_t ldx, const int32_t* ptr, float* out, int64_t ldo, uint32_t* mask,
                                    int64_t num_segments, int32_t F, int32_t mean, gml_stream_t stream) {
    if (num_segments < 0 || F != 32 || ldx < F || ldo < F) return F != 32 && F > 0 ? GML_E_UNSUPPORTED : GML_E_BADARG;
    if (num_segments == 0) return GML_OK;
    if (!x || !ptr || !out || !mask) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_segment_sum_mask32, dim3((unsigned)gml_cdiv(num_segments * 32, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, ptr, out, ldo, mask, num_segments, mean);
    return gml_launch_status();
}

// gradient of the pool for that layer, relu mask applied on the way: out[row][c] = g[seg[row]][c] * (c >= nrelu || bit c of mask[row]).
// One lane per (row, 4 columns): 4 + 4 bytes read (the gradient rows are cache hits), 128 written per row.
__global__ __launch_bounds__(256) void gml_k_segment_bcast_mask32(const float* __restrict__ g, int64_t ldg, const int32_t* __restrict__ seg,
                                                                  const uint32_t* __restrict__ mask, float* __restrict__ out, int64_t ldo,
                                                                  int64_t nrows, int nrelu) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = i >> 3;
    const int ch = (int)(i & 7);
    if (row >= nrows) return;
    const uint32_t m = mask[row] | (nrelu < 32 ? (~0u << nrelu) : 0u);
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + (int64_t)seg[row] * ldg + 4 * ch);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = ((m >> (4 * ch + k)) & 1u) ? v[k] : 0.f;
    *reinterpret_cast<f32x4*>(out + row * ldo + 4 * ch) = o;
}

extern "C" int gml_segment_bcast_mask(const float* g, int64_t ldg, const int32_t* seg, const uint32_t* mask, float* out, int64_t ldo,
                                      int64_t num_rows, int32_t F, int32_t nrelu, gml_stream_t stream) {
    if (num_rows < 0 || F != 32 || nrelu < 0 || nrelu > 32 || ldg < 32 || ldo < 32) return F != 32 && F > 0 ? GML_E_UNSUPPORTED : GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!g || !seg || !mask || !out || ((ldg | ldo) & 3) || (((uintptr_t)g | (uintptr_t)out) & 15)) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_segment_bcast_mask32, dim3((unsigned)gml_cdiv(num_rows * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       g, ldg, seg, mask, out, ldo, num_rows, nrelu);
    return gml_launch_status();
}

// gradient of the pooling: every row of a segment receives the segment's gradient row.  One wave per
// segment sweep: lanes <-> (row offset, column), coalesced stores.
__global__ void gml_k_segment_bcast(const float* __restrict__ g, int64_t ldg, const int32_t* __restrict__ ptr,
                                    float* __restrict__ out, int64_t ldo, int64_t nseg, int F, int mean) {
    const int64_t seg = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const int lane = threadIdx.x & 63;
    const int r0 = ptr[seg], r1 = ptr[seg + 1];
    const float sc = mean ? 1.f / (float)max(r1 - r0, 1) : 1.f;
    const int64_t n = (int64_t)(r1 - r0) * F;
    for (int64_t i = lane; i < n; i += 64) {
        const int64_t r = i / F;
        const int c = (int)(i - r * F);
        out[(r0 + r) * ldo + c] = g[seg * ldg + c] * sc;
    }
}

extern "C" int gml_segment_bcast(const float* g, int64_t ldg, const int32_t* ptr, float* out, int64_t ldo,
                                 int64_t num_segments, int32_t F, int32_t mean, gml_stream_t stream) {
    if (num_segments < 0 || F <= 0 || ldg < F || ldo < F) return GML_E_BADARG;
    if (num_segments == 0) return GML_OK;
    if (!g || !ptr || !out) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_segment_bcast, dim3((unsigned)gml_cdiv(num_segments, 4)), dim3(256), 0,
                       (hipStream_t)stream, g, ldg, ptr, out, ldo, num_segments, F, mean);
    return gml_launch_status();
}

// global_max_pool (torch_geometric.nn, /root/reference/enzymes.py:384): per segment and column the maximum over the
// segment's rows and the row that holds it (the FIRST one on ties: ascending scan with a strict compare); an empty
// segment gives 0 and argmax -1 (PyG's scatter-max fills empty outputs with 0).
__global__ void gml_k_segment_max(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
                                  float* __restrict__ out, int64_t ldo, int32_t* __restrict__ arg, int64_t nseg, int F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseg * F) return;
    const int64_t g = i / F;
    const int c = (int)(i % F);
    const int r0 = ptr[g], r1 = ptr[g + 1];
    float best = 0.f;
    int where = -1;
    for (int r = r0; r < r1; r += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[(int64_t)min(r + u, r1 - 1) * ldx + c];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r + u < r1 && (where < 0 || v[u] > best)) { best = v[u]; where = r + u; }
    }
    out[g * ldo + c] = best;
    if (arg) arg[g * F + c] = where;
}

extern "C" int gml_segment_max(const float* x, int64_t ldx, const int32_t* ptr, float* out, int64_t ldo, int32_t* argmax,
                               int64_t num_segments, int32_t F, gml_stream_t stream) {
    if (num_segments < 0 || F <= 0 || ldx < F || ldo < F) return GML_E_BADARG;
    if (num_segments == 0) return GML_OK;
    if (!x || !ptr || !out) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_segment_max, dim3((unsigned)gml_cdiv(num_segments * F, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, ptr, out, ldo, argmax, num_segments, F);
    return gml_launch_status();
}

// its gradient: dx[r, c] = g[seg, c] where r = argmax[seg, c], 0 elsewhere.  One wave per segment writes the segment's
// whole block (rows are owned by exactly one segment: no atomics, no separate zero fill).
__global__ void gml_k_segment_max_bwd(const float* __restrict__ g, int64_t ldg, const int32_t* __restrict__ ptr,
                                      const int32_t* __restrict__ arg, float* __restrict__ out, int64_t ldo, int64_t nseg, int F) {
    const int64_t seg = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const int lane = threadIdx.x & 63;
    const int r0 = ptr[seg], r1 = ptr[seg + 1];
    const int64_t n = (int64_t)(r1 - r0) * F;
    for (int64_t i = lane; i < n; i += 64) {
        const int64_t r = i / F;
        const int c = (int)(i - r * F);
        out[(r0 + r) * ldo + c] = (arg[seg * F + c] == r0 + (int)r) ? g[seg * ldg + c] : 0.f;
    }
}

extern "C" int gml_segment_max_bwd(const float* g, int64_t ldg, const int32_t* ptr, const int32_t* argmax, float* out,
                                   int64_t ldo, int64_t num_segments, int32_t F, gml_stream_t stream) {
    if (num_segments < 0 || F <= 0 || ldg < F || ldo < F) return GML_E_BADARG;
    if (num_segments == 0) return GML_OK;
    if (!g || !ptr || !argmax || !out) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_segment_max_bwd, dim3((unsigned)gml_cdiv(num_segments, 4)), dim3(256), 0,
                       (hipStream_t)stream, g, ldg, ptr, argmax, out, ldo, num_segments, F);
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

// =============================================================================================
// gml_fold_many: the partial-sum folds of several weight-gradient kernels in ONE launch (include/gml.h "Deferred folds").
// blockIdx.y = job, blockIdx.x = block of 16 columns; the same summation order as gml_k_reduce_rows / gml_k_reduce_partials /
// gml_k_split_fold (16 lanes over the partial index in steps of 16 through gml_fold_column, then the 16 sub-sums in ascending
// order): bit-identical to the per-kernel folds.
// =============================================================================================
struct GmlFoldJobs { gml_fold_job j[GML_FOLD_MAX_JOBS]; };

__global__ __launch_bounds__(256) void gml_k_fold_many(const GmlFoldJobs jobs) {
    __shared__ float red[16][17];
    const gml_fold_job& q = jobs.j[blockIdx.y];
    const int jl = threadIdx.x & 15, wl = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 16 + jl;
    if ((int64_t)blockIdx.x * 16 >= q.n) return;              // (uniform per block)
    float a = 0.f;
    if (j < q.n) a = gml_fold_column(q.partial, q.nparts, q.n, j, wl);
    red[wl][jl] = a;
    __syncthreads();
    if (wl != 0 || j >= q.n) return;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][jl];
    int64_t off = j;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (off < q.ndst[k]) { if (q.dst[k]) q.dst[k][off] = t; return; }
        off -= q.ndst[k];
    }
}

extern "C" int gml_fold_many(const gml_fold_job* jobs, int32_t njobs, gml_stream_t stream) {
    if (njobs < 0 || njobs > GML_FOLD_MAX_JOBS || (njobs > 0 && !jobs)) return GML_E_BADARG;
    if (njobs == 0) return GML_OK;
    GmlFoldJobs a = {};
    int64_t nmax = 0;
    for (int i = 0; i < njobs; ++i) {
        const gml_fold_job& q = jobs[i];
        int64_t tot = 0;
        for (int k = 0; k < 5; ++k) { if (q.ndst[k] < 0) return GML_E_BADARG; tot += q.ndst[k]; }
        if (q.n < 0 || q.nparts < 0 || tot > q.n || (q.n > 0 && q.nparts > 0 && !q.partial)) return GML_E_BADARG;
        a.j[i] = q;
        nmax = q.n > nmax ? q.n : nmax;
    }
    if (nmax == 0) return GML_OK;
    hipLaunchKernelGGL(gml_k_fold_many, dim3((unsigned)gml_cdiv(nmax, 16), (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, a);
    return gml_launch_status();
}

// =============================================================================================
// Adam over a list of parameter tensors in ONE launch (torch.optim.Adam's update, Zinc12k.py:349: no weight decay, no amsgrad):
//   t = step[0] + 1;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// step [1] (fp32, on the device: the launch is capturable in a HIP graph and every replay advances it) is read by every workgroup
// at its start and written back as t by the LAST workgroup to finish (done [1], uint32, zero between launches), i.e. after every
// other workgroup has read it.  torch's fused Adam is two multi-tensor launches for ZINC's 44 tensors plus one for the step counts: at
// the reference's batch size (a step = ~30 launches of microseconds each) that is 17 of a step's 270 us.
// =============================================================================================
struct GmlAdamJobs { gml_adam_job j[GML_ADAM_MAX_JOBS]; int32_t first_block[GML_ADAM_MAX_JOBS + 1]; };

__device__ __forceinline__ void gml_adam_elem(const gml_adam_job& q, int64_t i, float b1, float b2, float ss, float rc2, float eps) {
    const float g = q.g[i];
    const float m = fmaf(b1, q.m[i], (1.f - b1) * g);
    const float v = fmaf(b2, q.v[i], (1.f - b2) * g * g);
    q.m[i] = m; q.v[i] = v;
    q.p[i] -= ss * m / fmaf(sqrtf(v), rc2, eps);
}

// 4096 elements per workgroup, workgroups laid out job after job (first_block: ZINC's 33 k parameters in 44 tensors are 52 workgroups
// -- one per 1024 elements on a [chunks of the largest tensor] x [tensors] grid was 352, and their 352 same-address atomics cost more
// than the update; ONE workgroup walking everything is a chain of 33 dependent memory round trips per thread: 40 us).
__global__ __launch_bounds__(256) void gml_k_adam_many(const GmlAdamJobs jobs, int njobs, float* step, unsigned* done, float lr, float b1,
                                                       float b2, float eps) {
    // step count hand-shake at the START of the workgroup, off the update's critical path: every thread reads step[0]; behind a barrier
    // thread 0 counts the workgroup in; the LAST workgroup to be counted writes the new count (every other one has read the old one by
    // then and carries it in registers) and clears the counter for the next launch
    const float t = __builtin_nontemporal_load(step) + 1.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(done, 1u) == gridDim.x - 1) { *done = 0u; step[0] = t; }
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs.first_block[j + 1]) ++j;
    const gml_adam_job& q = jobs.j[j];
    // 1 - b^t without the cancellation of 1 - pow(b, t) at small t (b2 = 0.999, t = 1: 4e-5 relative in fp32 with the fast pow --
    // torch computes the corrections on the host in double): -expm1(t log(b)), log(b) = log1p(b - 1); once per thread
    const float c1 = -expm1f(t * log1pf(b1 - 1.f)), c2 = -expm1f(t * log1pf(b2 - 1.f));
    const float ss = lr / c1, rc2 = 1.f / sqrtf(c2);
    const int64_t base = (int64_t)((int)blockIdx.x - jobs.first_block[j]) * 4096;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i < q.n) gml_adam_elem(q, i, b1, b2, ss, rc2, eps);
    }
}

extern "C" int gml_adam_many(const gml_adam_job* jobs, int32_t njobs, float* step, uint32_t* done, float lr, float beta1, float beta2,
                             float eps, gml_stream_t stream) {
    if (njobs < 0 || njobs > GML_ADAM_MAX_JOBS || (njobs > 0 && !jobs) || !step || !done) return GML_E_BADARG;
    if (njobs == 0) return GML_OK;
    GmlAdamJobs a = {};
    int64_t total = 0, blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const gml_adam_job& q = jobs[i];
        if (q.n < 0 || (q.n > 0 && (!q.p || !q.g || !q.m || !q.v))) return GML_E_BADARG;
        a.j[i] = q;
        a.first_block[i] = (int32_t)blocks;
        blocks += gml_cdiv(q.n, 4096);
        total += q.n;
        if (blocks > 0x7fffffff) return GML_E_UNSUPPORTED;
    }
    a.first_block[njobs] = (int32_t)blocks;
    if (total == 0) return GML_OK;
    hipLaunchKernelGGL(gml_k_adam_many, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, njobs, step, done, lr, beta1, beta2, eps);
    return gml_launch_status();
}
