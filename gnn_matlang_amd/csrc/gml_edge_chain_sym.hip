// C-ABI of the edge branch over a batch's UNIQUE support rows (gml_edge_chain_sym_impl.h): the pairing pass, the forward of a layer
// stack (three-piece products; stacks: S in {4, 8}, single layers: 2 <= S <= 16) and the backward (two-piece chains), 2 <= S = Sout <= 16
#include "gml_edge_chain_sym_impl.h"
#include "gml_edge_chain16_impl.h"
#include "gml_edge_chain16x6_impl.h"

extern "C" int gml_edge_sym_flags(const int32_t* rowptr_t, const int32_t* col_t, const float* val_s, int64_t num_rows,
                                  int64_t num_edges, int32_t S, int32_t* flag, int32_t* mirror, gml_stream_t stream) {
    if (num_rows < 0 || num_edges < 0 || S <= 0) return GML_E_BADARG;
    if (num_edges == 0) return GML_OK;
    if (!rowptr_t || !col_t || !val_s || !flag || !mirror || num_rows == 0) return GML_E_BADARG;
    // lanes per source row: the power of two nearest the mean row length (GML_SYM_LPS in the environment overrides: A/B)
    static const int lps_env = [] { const char* e = getenv("GML_SYM_LPS"); return e ? atoi(e) : -1; }();
    int lps = 0;
    while (lps < 4 && (num_rows << (lps + 1)) <= num_edges) ++lps;        // 2^lps <= edges per row
    if (lps_env >= 0 && lps_env <= 4) lps = lps_env;
    hipLaunchKernelGGL(gml_k_edge_sym_flags, dim3((unsigned)gml_cdiv(num_rows << lps, 256)), dim3(256), 0, (hipStream_t)stream, rowptr_t,
                       col_t, reinterpret_cast<const uint32_t*>(val_s), num_rows, num_edges, (int)S, lps, flag, mirror);
    return gml_launch_status();
}

template <int S, int L>
static int sym_fwd_go(const float* ea, const int32_t* uid, const int32_t* mir, int64_t U, const float* const* w1, const float* const* w2,
                      const float* const* w3, const float* const* w4, float* const* out, int64_t E, hipStream_t st) {
    GmlChain6Stack<L> a;
    for (int l = 0; l < L; ++l) { a.w1[l] = w1[l]; a.w2[l] = w2[l]; a.w3[l] = w3[l]; a.w4[l] = w4[l]; a.out[l] = out[l]; }
    return gml_launch_edge_chain6_fwd_sym<S, L>(ea, uid, mir, a, E, U, st);
}

extern "C" int gml_edge_mlp_fwd_stack6_sym(const float* ea, const int32_t* uid, const int32_t* mir, int64_t num_unique, int32_t nlayers,
                                           const float* const* w1, const float* const* w2, const float* const* w3,
                                           const float* const* w4, float* const* out, int64_t num_edges, int32_t S, int32_t Sout,
                                           gml_stream_t stream) {
    if (num_edges < 0 || num_unique < 0 || num_unique > num_edges || S <= 0 || Sout <= 0 || nlayers <= 0 || !w1 || !w2 || !w3 || !w4 || !out)
        return GML_E_BADARG;
    if (S != Sout || S < 2 || S > 16 || nlayers > 4 || (S != 8 && S != 4 && nlayers > 1)) return GML_E_UNSUPPORTED;
    if ((uint64_t)num_edges * (uint64_t)S * 4u >= 0x7fffff00ull) return GML_E_UNSUPPORTED;      /* 32-bit store offsets */
    if (num_edges == 0) return GML_OK;
    if (num_unique == 0 || !ea || !uid || !mir || (S % 4 == 0 && (((uintptr_t)ea) & 15) != 0)) return GML_E_BADARG;
    for (int l = 0; l < nlayers; ++l)
        if (!w1[l] || !w2[l] || !w3[l] || !w4[l] || !out[l] || (S % 4 == 0 && (((uintptr_t)out[l]) & 15) != 0)) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
#define GML_SYM_GO(SV, LV) if (S == SV && nlayers == LV) return sym_fwd_go<SV, LV>(ea, uid, mir, num_unique, w1, w2, w3, w4, out, num_edges, st);
    GML_SYM_GO(8, 1) GML_SYM_GO(8, 2) GML_SYM_GO(8, 3) GML_SYM_GO(8, 4)
    GML_SYM_GO(4, 1) GML_SYM_GO(4, 2) GML_SYM_GO(4, 3) GML_SYM_GO(4, 4)
    GML_SYM_GO(2, 1) GML_SYM_GO(3, 1) GML_SYM_GO(5, 1) GML_SYM_GO(6, 1) GML_SYM_GO(7, 1)
#define GML_SYM16_GO(SV) if (S == SV) return gml_launch_edge_chain16x6_fwd_sym<SV>(ea, uid, mir, num_unique, w1[0], w2[0], w3[0], w4[0], out[0], st);
    GML_SYM16_GO(9) GML_SYM16_GO(10) GML_SYM16_GO(11) GML_SYM16_GO(12) GML_SYM16_GO(13) GML_SYM16_GO(14) GML_SYM16_GO(15) GML_SYM16_GO(16)
    return GML_E_UNSUPPORTED;
}

extern "C" int64_t gml_edge_mlp_bwd_sym_parts(int64_t num_unique, int32_t S) {
    if (num_unique <= 0) return 0;
    return S > 8 ? gml_edge_chain16_bwd_groups(num_unique) : gml_edge_chain_bwd_groups(num_unique, gml_edge_chain_bwd_wgs());
}

template <int S>
static int sym_bwd16_go(const uint32_t* es, const int32_t* uid, const int32_t* mir, int64_t U, const float* w1, const float* w2,
                        const float* w3, const float* w4, const float* gout, float* dw1, float* dw2, float* dw3, float* dw4, void* ws,
                        size_t ws_bytes, hipStream_t st) {
    const int64_t ntiles = gml_cdiv(U, 16);
    const int64_t grid = gml_edge_chain16_bwd_groups(U);
    constexpr int NW = GML_CHAIN16_NW(S);
    if (ws_bytes < (size_t)grid * NW * sizeof(float)) return GML_E_WORKSPACE;
    hipLaunchKernelGGL((gml_k_edge_chain16_bwd<S, true>), dim3((unsigned)grid), dim3(256), 0, st, es, w1, w2, w3, w4, gout, (float*)ws, U, ntiles,
                       uid, mir);
    int rc = gml_launch_status();
    if (rc != GML_OK || !dw1) return rc;
    const int n123 = 2 * S * S, n4 = S * 4 * S;
    hipLaunchKernelGGL(gml_k_reduce_partials, dim3((unsigned)gml_cdiv(NW, 16)), dim3(256), 0, st, (const float*)ws, grid, NW, dw1, n123,
                       dw2, n123, dw3, n123, dw4, n4);
    return gml_launch_status();
}

template <int S>
static int sym_bwd_go(const uint32_t* es, const int32_t* uid, const int32_t* mir, int64_t U, const float* w1, const float* w2,
                      const float* w3, const float* w4, const float* gout, float* dw1, float* dw2, float* dw3, float* dw4, void* ws,
                      size_t ws_bytes, hipStream_t st) {
    const int64_t ntiles = gml_cdiv(U, 16);
    const int64_t grid = gml_edge_chain_bwd_groups(U, gml_edge_chain_bwd_wgs());
    constexpr int NW = GML_CHAIN_NW(S);
    if (ws_bytes < (size_t)grid * NW * sizeof(float)) return GML_E_WORKSPACE;
    hipLaunchKernelGGL((gml_k_edge_chain_bwd_sym<S>), dim3((unsigned)grid), dim3(256), 0, st, es, uid, mir, w1, w2, w3, w4, gout,
                       (float*)ws, U, ntiles);
    int rc = gml_launch_status();
    if (rc != GML_OK || !dw1) return rc;                     /* dw1 == NULL: the partials stay in ws (gml_fold_many) */
    const int n123 = 2 * S * S, n4 = S * 4 * S;
    hipLaunchKernelGGL(gml_k_reduce_partials, dim3((unsigned)gml_cdiv(NW, 16)), dim3(256), 0, st, (const float*)ws, grid, NW, dw1, n123,
                       dw2, n123, dw3, n123, dw4, n4);
    return gml_launch_status();
}

extern "C" int gml_edge_mlp_bwd_sym(const void* ea_split, const int32_t* uid, const int32_t* mir, int64_t num_unique, const float* w1,
                                    const float* w2, const float* w3, const float* w4, const float* gout, float* dw1, float* dw2,
                                    float* dw3, float* dw4, int64_t num_edges, int32_t S, int32_t Sout, void* ws, size_t ws_bytes,
                                    gml_stream_t stream) {
    if (num_edges <= 0 || num_unique <= 0 || num_unique > num_edges || S <= 0 || Sout <= 0) return GML_E_BADARG;
    const bool nofold = !dw1 && !dw2 && !dw3 && !dw4;
    if (!w1 || !w2 || !w3 || !w4 || (!nofold && (!dw1 || !dw2 || !dw3 || !dw4))) return GML_E_BADARG;
    if (S != Sout || S < 2 || S > 16) return GML_E_UNSUPPORTED;
    if (!ea_split || !uid || !mir || !gout || !ws || (((uintptr_t)ea_split) & 15) != 0 || (S % 4 == 0 && (((uintptr_t)gout) & 15) != 0)) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t* es = (const uint32_t*)ea_split;
    switch (S) {
#define GML_SYM_BWD(SV) case SV: return sym_bwd_go<SV>(es, uid, mir, num_unique, w1, w2, w3, w4, gout, dw1, dw2, dw3, dw4, ws, ws_bytes, st);
        GML_SYM_BWD(2) GML_SYM_BWD(3) GML_SYM_BWD(4) GML_SYM_BWD(5) GML_SYM_BWD(6) GML_SYM_BWD(7) GML_SYM_BWD(8)
#define GML_SYM_BWD16(SV) case SV: return sym_bwd16_go<SV>(es, uid, mir, num_unique, w1, w2, w3, w4, gout, dw1, dw2, dw3, dw4, ws, ws_bytes, st);
        GML_SYM_BWD16(9) GML_SYM_BWD16(10) GML_SYM_BWD16(11) GML_SYM_BWD16(12) GML_SYM_BWD16(13) GML_SYM_BWD16(14) GML_SYM_BWD16(15) GML_SYM_BWD16(16)
    }
    return GML_E_UNSUPPORTED;
}
