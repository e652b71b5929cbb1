// C-ABI dispatch of the ML3Layer edge-branch kernels + the partial-sum fold.
#include "gml_edge_mlp_impl.h"
#include "gml_edge_chain_impl.h"
#include "gml_edge_chain16_impl.h"
#include <stdlib.h>

// 2 <= S <= 8 runs on the bf16 matrix cores (gml_edge_chain_impl.h); GML_EDGE_VALU=1 in the environment keeps the
// one-edge-per-lane fp32 VALU kernels for every S (ablation / exact-fp32 arithmetic).  S = 1 stays on the VALU
// kernels: its contractions are single products, so the split's 2^-17 rounding is not averaged over a sum (measured
// 5e-5 .. 1e-4 of the output scale against 1e-5 for S >= 2), and there is no arithmetic to save.
static bool emlp_use_chain(int S) {
    static const bool valu = [] { const char* e = getenv("GML_EDGE_VALU"); return e && e[0] == '1'; }();
    return S >= 2 && S <= 8 && !valu;
}
#define GML_DECL_ECHAIN(SV)                                                                                  \
    template <> int gml_launch_edge_chain_fwd<SV>(const float*, const uint32_t*, const float*, const float*, \
                                                  const float*, const float*, float*, const int32_t*, float*, \
                                                  int64_t, hipStream_t);                                     \
    template <> int gml_launch_edge_chain_bwd<SV>(const float*, const uint32_t*, const float*, const float*, \
                                                  const float*, const float*, const float*, float*, float*,  \
                                                  float*, float*, float*, int64_t, void*, size_t, hipStream_t);
GML_DECL_ECHAIN(1) GML_DECL_ECHAIN(2) GML_DECL_ECHAIN(3) GML_DECL_ECHAIN(4)
GML_DECL_ECHAIN(5) GML_DECL_ECHAIN(6) GML_DECL_ECHAIN(7) GML_DECL_ECHAIN(8)
// 8 < S <= 16 (counting.py: S = 12): the K = 16-slot chain of gml_edge_chain16_impl.h; needs the 64-byte pre-split rows and
// produces no gradient for the raw supports (those cases stay on the VALU kernels)
static bool emlp_use_chain16(int S, const void* ea_split, const void* gin) {
    static const bool valu = [] { const char* e = getenv("GML_EDGE_VALU"); return e && e[0] == '1'; }();
    return S > 8 && S <= 16 && !valu && ea_split != nullptr && gin == nullptr;
}
#define GML_DECL_ECHAIN16(SV)                                                                                \
    template <> int gml_launch_edge_chain16_fwd<SV>(const uint32_t*, const float*, const float*, const float*, \
                                                    const float*, float*, const int32_t*, float*, int64_t, hipStream_t); \
    template <> int gml_launch_edge_chain16_bwd<SV>(const uint32_t*, const float*, const float*, const float*, \
                                                    const float*, const float*, float*, float*, float*, float*, \
                                                    int64_t, void*, size_t, hipStream_t);
GML_DECL_ECHAIN16(9) GML_DECL_ECHAIN16(10) GML_DECL_ECHAIN16(11) GML_DECL_ECHAIN16(12)
GML_DECL_ECHAIN16(13) GML_DECL_ECHAIN16(14) GML_DECL_ECHAIN16(15) GML_DECL_ECHAIN16(16)
#define GML_ECHAIN16_SWITCH(CALL)                                                               \
    switch (S) {                                                                                \
        case 9: return CALL(9); case 10: return CALL(10); case 11: return CALL(11);             \
        case 12: return CALL(12); case 13: return CALL(13); case 14: return CALL(14);           \
        case 15: return CALL(15); case 16: return CALL(16);                                     \
    }
#define GML_ECHAIN_SWITCH(CALL)                                                                 \
    switch (S) {                                                                                \
        case 1: return CALL(1); case 2: return CALL(2); case 3: return CALL(3);                 \
        case 4: return CALL(4); case 5: return CALL(5); case 6: return CALL(6);                 \
        case 7: return CALL(7); case 8: return CALL(8);                                         \
    }

// dst[j] = sum_w partial[w][j] in a fixed order: 16 lanes split the partial index, LDS tree in fixed order
__global__ __launch_bounds__(256) void gml_k_reduce_partials(const float* __restrict__ partial, int64_t nwaves, int nw,
                                                            float* __restrict__ d0, int n0, float* __restrict__ d1, int n1,
                                                            float* __restrict__ d2, int n2, float* __restrict__ d3, int n3) {
    __shared__ float red[16][17];
    const int jl = threadIdx.x & 15, wl = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + jl;
    float a = 0.f;
    if (j < nw) a = gml_fold_column(partial, nwaves, nw, j, wl);
    red[wl][jl] = a;
    __syncthreads();
    if (wl == 0 && j < nw) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][jl];
        if (j < n0) d0[j] = t;
        else if (j < n0 + n1) d1[j - n0] = t;
        else if (j < n0 + n1 + n2) d2[j - n0 - n1] = t;
        else if (j < n0 + n1 + n2 + n3) d3[j - n0 - n1 - n2] = t;
    }
}

#define GML_DECL_EMLP(SV)                                                                                    \
    template <> int gml_launch_edge_mlp_fwd<SV, SV>(const float*, const float*, const float*, const float*,  \
                                                    const float*, float*, const int32_t*, float*, int64_t,   \
                                                    hipStream_t);                                            \
    template <> int gml_launch_edge_mlp_bwd<SV, SV>(const float*, const float*, const float*, const float*,  \
                                                    const float*, const float*, float*, float*, float*,      \
                                                    float*, float*, int64_t, void*, size_t, hipStream_t);
GML_DECL_EMLP(1) GML_DECL_EMLP(2) GML_DECL_EMLP(3) GML_DECL_EMLP(4) GML_DECL_EMLP(5) GML_DECL_EMLP(6)
GML_DECL_EMLP(7) GML_DECL_EMLP(8) GML_DECL_EMLP(9) GML_DECL_EMLP(10) GML_DECL_EMLP(11) GML_DECL_EMLP(12)
GML_DECL_EMLP(13) GML_DECL_EMLP(14) GML_DECL_EMLP(15) GML_DECL_EMLP(16)

#define GML_EMLP_SWITCH(CALL)                                                                   \
    switch (S) {                                                                                \
        case 1: return CALL(1); case 2: return CALL(2); case 3: return CALL(3);                 \
        case 4: return CALL(4); case 5: return CALL(5); case 6: return CALL(6);                 \
        case 7: return CALL(7); case 8: return CALL(8); case 9: return CALL(9);                 \
        case 10: return CALL(10); case 11: return CALL(11); case 12: return CALL(12);           \
        case 13: return CALL(13); case 14: return CALL(14); case 15: return CALL(15);           \
        case 16: return CALL(16);                                                               \
    }                                                                                           \
    return GML_E_UNSUPPORTED;

// hi[8] | lo[8] bf16 per edge (32 bytes): the layer-1 operand of the matrix-core kernels, made once per batch
// 8 < S <= 16: hi[16] | lo[16] per edge (64 bytes), the operand rows of gml_edge_chain16_impl.h
__global__ __launch_bounds__(256) void gml_k_edge_presplit16(const float* __restrict__ ea, uint32_t* __restrict__ es,
                                                            int64_t E, int S) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    uint32_t hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x0 = (2 * j < S) ? ea[e * S + 2 * j] : 0.f, x1 = (2 * j + 1 < S) ? ea[e * S + 2 * j + 1] : 0.f;
        const float t0 = __uint_as_float(__float_as_uint(x0) & 0xffff0000u);
        const float t1 = __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        hi[j] = gml_pack2(t0, t1);
        lo[j] = gml_pack2(x0 - t0, x1 - t1);
    }
    u32x4* o = reinterpret_cast<u32x4*>(es + e * 16);
    o[0] = u32x4{hi[0], hi[1], hi[2], hi[3]};
    o[1] = u32x4{hi[4], hi[5], hi[6], hi[7]};
    o[2] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    o[3] = u32x4{lo[4], lo[5], lo[6], lo[7]};
}

__global__ __launch_bounds__(256) void gml_k_edge_presplit(const float* __restrict__ ea, uint32_t* __restrict__ es,
                                                          int64_t E, int S) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x0 = (2 * j < S) ? ea[e * S + 2 * j] : 0.f, x1 = (2 * j + 1 < S) ? ea[e * S + 2 * j + 1] : 0.f;
        const float t0 = __uint_as_float(__float_as_uint(x0) & 0xffff0000u);
        const float t1 = __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        hi[j] = gml_pack2(t0, t1);
        lo[j] = gml_pack2(x0 - t0, x1 - t1);
    }
    u32x4* o = reinterpret_cast<u32x4*>(es + e * 8);
    o[0] = u32x4{hi[0], hi[1], hi[2], hi[3]};
    o[1] = u32x4{lo[0], lo[1], lo[2], lo[3]};
}

// the same with the row gather of the value sort in front: out[k] = in[perm[k]] and its pre-split in ONE pass (a fresh
// batch otherwise reads and writes the supports twice: gml_gather_rows, then gml_edge_presplit)
__global__ __launch_bounds__(256) void gml_k_gather_presplit(const float* __restrict__ in, const int32_t* __restrict__ perm,
                                                            float* __restrict__ out, uint32_t* __restrict__ es,
                                                            int64_t E, int S) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const float* src = in + (int64_t)perm[e] * S;
    float x[8];
    if (S == 8) {
        const f32x4 a = reinterpret_cast<const f32x4*>(src)[0], b = reinterpret_cast<const f32x4*>(src)[1];
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        reinterpret_cast<f32x4*>(out + e * 8)[0] = a;
        reinterpret_cast<f32x4*>(out + e * 8)[1] = b;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = (j < S) ? src[j] : 0.f;
            if (j < S) out[e * S + j] = x[j];
        }
    }
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t0 = __uint_as_float(__float_as_uint(x[2 * j]) & 0xffff0000u);
        const float t1 = __uint_as_float(__float_as_uint(x[2 * j + 1]) & 0xffff0000u);
        hi[j] = gml_pack2(t0, t1);
        lo[j] = gml_pack2(x[2 * j] - t0, x[2 * j + 1] - t1);
    }
    u32x4* o = reinterpret_cast<u32x4*>(es + e * 8);
    o[0] = u32x4{hi[0], hi[1], hi[2], hi[3]};
    o[1] = u32x4{lo[0], lo[1], lo[2], lo[3]};
}

extern "C" int gml_gather_rows_presplit(const float* in, const int32_t* perm, float* out, void* out_split, int64_t rows,
                                        int32_t S, gml_stream_t stream) {
    if (rows < 0 || S <= 0) return GML_E_BADARG;
    if (S > 8) return GML_E_UNSUPPORTED;
    if (rows == 0) return GML_OK;
    if (!in || !perm || !out || !out_split) return GML_E_BADARG;
    if ((((uintptr_t)out_split) & 15) != 0 || (S == 8 && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) != 0)) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_gather_presplit, dim3((unsigned)gml_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, in, perm,
                       out, (uint32_t*)out_split, rows, S);
    return gml_launch_status();
}

extern "C" int gml_edge_presplit(const float* ea, void* ea_split, int64_t num_edges, int32_t S, gml_stream_t stream) {
    if (num_edges < 0 || S <= 0) return GML_E_BADARG;
    if (S > 16) return GML_E_UNSUPPORTED;
    if (num_edges == 0) return GML_OK;
    if (!ea || !ea_split || (((uintptr_t)ea_split) & 15) != 0) return GML_E_BADARG;
    if (S > 8) {
        hipLaunchKernelGGL(gml_k_edge_presplit16, dim3((unsigned)gml_cdiv(num_edges, 256)), dim3(256), 0, (hipStream_t)stream,
                           ea, (uint32_t*)ea_split, num_edges, S);
        return gml_launch_status();
    }
    hipLaunchKernelGGL(gml_k_edge_presplit, dim3((unsigned)gml_cdiv(num_edges, 256)), dim3(256), 0, (hipStream_t)stream, ea,
                       (uint32_t*)ea_split, num_edges, S);
    return gml_launch_status();
}

extern "C" int gml_edge_mlp_fwd(const float* ea, const void* ea_split, const float* w1, const float* w2, const float* w3,
                                const float* w4, float* out, const int32_t* tpos, float* out_t,
                                int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream) {
    if (num_edges < 0 || S <= 0 || Sout <= 0) return GML_E_BADARG;
    if (num_edges == 0) return GML_OK;
    if (!ea || !w1 || !w2 || !w3 || !w4 || !out) return GML_E_BADARG;
    if (S != Sout) return GML_E_UNSUPPORTED;   // every reference script uses nedgeoutput == nedgeinput
    if ((((uintptr_t)ea | (uintptr_t)out | (uintptr_t)out_t | (uintptr_t)ea_split) & 15) != 0) return GML_E_BADARG;
#ifdef GML_NO_PRESPLIT
    ea_split = nullptr;
#endif
    if (out_t && !tpos) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (emlp_use_chain(S)) {
        // the second (source-order) copy is scattered through one buffer descriptor: 32-bit byte offsets
        if (out_t && (uint64_t)num_edges * (uint64_t)S * 4u >= 0xffffff00ull) return GML_E_UNSUPPORTED;
#define GML_CALL_CF(SV) \
    gml_launch_edge_chain_fwd<SV>(ea, (const uint32_t*)ea_split, w1, w2, w3, w4, out, tpos, out_t, num_edges, st)
        GML_ECHAIN_SWITCH(GML_CALL_CF)
    }
    if (emlp_use_chain16(S, ea_split, nullptr)) {
#define GML_CALL_CF16(SV) \
    gml_launch_edge_chain16_fwd<SV>((const uint32_t*)ea_split, w1, w2, w3, w4, out, tpos, out_t, num_edges, st)
        GML_ECHAIN16_SWITCH(GML_CALL_CF16)
    }
#define GML_CALL_F(SV) gml_launch_edge_mlp_fwd<SV, SV>(ea, w1, w2, w3, w4, out, tpos, out_t, num_edges, st)
    GML_EMLP_SWITCH(GML_CALL_F)
}

// the same on the exact-arithmetic family whatever the shape (one edge per lane, fp32 FMAs, f32-input MFMA for the weight gradients, the
// library's tanh): what GML_F32_MFMA is to the conv kernels.  The matrix-core chains split their operands into bf16 pairs and use a
// short tanh (~2e-7 absolute): 5e-7 rms on the learned supports where this family -- like torch's fp32 on the CPU -- carries ~1e-8.
extern "C" int gml_edge_mlp_fwd_exact(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, float* out,
                                      const int32_t* tpos, float* out_t, int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream) {
    if (num_edges < 0 || S <= 0 || Sout <= 0) return GML_E_BADARG;
    if (num_edges == 0) return GML_OK;
    if (!ea || !w1 || !w2 || !w3 || !w4 || !out) return GML_E_BADARG;
    if (S != Sout) return GML_E_UNSUPPORTED;
    if ((((uintptr_t)ea | (uintptr_t)out | (uintptr_t)out_t) & 15) != 0) return GML_E_BADARG;
    if (out_t && !tpos) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    GML_EMLP_SWITCH(GML_CALL_F)
}

static int64_t emlp_bwd_waves(int64_t E, int S) {
    const int chb = 7 * S, cha = 5 * S;
    const int str = (chb > cha ? chb : cha) | 1;
    const int waves = (str * 64 * 4 * 4 <= 64 * 1024) ? 4 : 2;
    return gml_edge_mlp_bwd_waves(E, waves);
}

extern "C" size_t gml_edge_mlp_bwd_workspace_bytes(int64_t num_edges, int32_t S, int32_t Sout) {
    if (num_edges <= 0 || S <= 0 || Sout != S) return 0;
    // both kernel families are covered, so the size does not depend on the environment switch
    int64_t parts = emlp_bwd_waves(num_edges, S);
    if (S <= 8 && gml_edge_chain_bwd_groups(num_edges) > parts) parts = gml_edge_chain_bwd_groups(num_edges);
    if (S > 8 && S <= 16 && gml_edge_chain16_bwd_groups(num_edges) > parts) parts = gml_edge_chain16_bwd_groups(num_edges);
    return (size_t)parts * (size_t)(6 * S * S + Sout * 4 * S) * sizeof(float);
}

// partial rows gml_edge_mlp_bwd leaves in ws for this call shape (the dispatch below, mirrored)
extern "C" int64_t gml_edge_mlp_bwd_parts(int64_t num_edges, int32_t S, int32_t Sout, int32_t has_split, int32_t want_gin) {
    if (num_edges <= 0 || S <= 0 || Sout != S) return 0;
#ifdef GML_NO_PRESPLIT
    has_split = 0;
#endif
    if (emlp_use_chain(S)) return gml_edge_chain_bwd_groups(num_edges, gml_edge_chain_bwd_wgs());
    if (emlp_use_chain16(S, has_split ? (const void*)1 : nullptr, want_gin ? (const void*)1 : nullptr)) return gml_edge_chain16_bwd_groups(num_edges);
    return emlp_bwd_waves(num_edges, S);
}

extern "C" int gml_edge_mlp_bwd(const float* ea, const void* ea_split, const float* w1, const float* w2, const float* w3,
                                const float* w4, const float* gout, float* gin, float* dw1, float* dw2,
                                float* dw3, float* dw4, int64_t num_edges, int32_t S, int32_t Sout,
                                void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (num_edges < 0 || S <= 0 || Sout <= 0) return GML_E_BADARG;
    const bool nofold = !dw1 && !dw2 && !dw3 && !dw4;        /* all four NULL: the partials stay in ws (gml_fold_many, gml_edge_mlp_bwd_parts) */
    if (!w1 || !w2 || !w3 || !w4 || (!nofold && (!dw1 || !dw2 || !dw3 || !dw4))) return GML_E_BADARG;
    if (S != Sout) return GML_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (num_edges == 0) {
        if (nofold) return GML_OK;                           /* (gml_edge_mlp_bwd_parts is 0: the fold writes zeros) */
        gml_zero_async(dw1, sizeof(float) * 2 * S * S, st);
        gml_zero_async(dw2, sizeof(float) * 2 * S * S, st);
        gml_zero_async(dw3, sizeof(float) * 2 * S * S, st);
        gml_zero_async(dw4, sizeof(float) * 4 * S * Sout, st);
        return gml_launch_status();
    }
    if (!ea || !gout || !ws) return GML_E_BADARG;
    if ((((uintptr_t)ea | (uintptr_t)gout | (uintptr_t)gin | (uintptr_t)ea_split) & 15) != 0) return GML_E_BADARG;
#ifdef GML_NO_PRESPLIT
    ea_split = nullptr;
#endif
    if (emlp_use_chain(S)) {
#define GML_CALL_CB(SV) \
    gml_launch_edge_chain_bwd<SV>(ea, (const uint32_t*)ea_split, w1, w2, w3, w4, gout, gin, dw1, dw2, dw3, dw4, \
                                  num_edges, ws, ws_bytes, st)
        GML_ECHAIN_SWITCH(GML_CALL_CB)
    }
    if (emlp_use_chain16(S, ea_split, gin)) {
#define GML_CALL_CB16(SV) \
    gml_launch_edge_chain16_bwd<SV>((const uint32_t*)ea_split, w1, w2, w3, w4, gout, dw1, dw2, dw3, dw4, num_edges, ws, \
                                    ws_bytes, st)
        GML_ECHAIN16_SWITCH(GML_CALL_CB16)
    }
#define GML_CALL_B(SV) \
    gml_launch_edge_mlp_bwd<SV, SV>(ea, w1, w2, w3, w4, gout, gin, dw1, dw2, dw3, dw4, num_edges, ws, ws_bytes, st)
    GML_EMLP_SWITCH(GML_CALL_B)
}

// gml_edge_mlp_bwd on the exact-arithmetic family (see gml_edge_mlp_fwd_exact); dw1 .. dw4 are required (no deferred fold);
// ws: gml_edge_mlp_bwd_workspace_bytes (it covers both families)
extern "C" int gml_edge_mlp_bwd_exact(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4,
                                      const float* gout, float* gin, float* dw1, float* dw2, float* dw3, float* dw4,
                                      int64_t num_edges, int32_t S, int32_t Sout, void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (num_edges < 0 || S <= 0 || Sout <= 0) return GML_E_BADARG;
    if (!w1 || !w2 || !w3 || !w4 || !dw1 || !dw2 || !dw3 || !dw4) return GML_E_BADARG;
    if (S != Sout) return GML_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (num_edges == 0) {
        gml_zero_async(dw1, sizeof(float) * 2 * S * S, st);
        gml_zero_async(dw2, sizeof(float) * 2 * S * S, st);
        gml_zero_async(dw3, sizeof(float) * 2 * S * S, st);
        gml_zero_async(dw4, sizeof(float) * 4 * S * Sout, st);
        return gml_launch_status();
    }
    if (!ea || !gout || !ws) return GML_E_BADARG;
    if ((((uintptr_t)ea | (uintptr_t)gout | (uintptr_t)gin) & 15) != 0) return GML_E_BADARG;
    GML_EMLP_SWITCH(GML_CALL_B)
}
