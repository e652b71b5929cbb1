// gml_edge_mlp_fwd6: one layer's edge branch, three-piece products: 2 <= S = Sout <= 8 (gml_edge_chain6_impl.h), 9 .. 16 (gml_edge_chain16x6_impl.h)
#include "gml_edge_chain6_impl.h"
#include "gml_edge_chain16x6_impl.h"

template <int S>
static int one_go(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, float* out, const int32_t* tpos,
                  float* out_t, int64_t E, hipStream_t st) {
    GmlChain6Stack<1> a;
    a.w1[0] = w1; a.w2[0] = w2; a.w3[0] = w3; a.w4[0] = w4; a.out[0] = out;
    return gml_launch_edge_chain6_fwd<S, 1>(ea, a, tpos, out_t, E, st);
}

extern "C" int gml_edge_mlp_fwd6(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, float* out,
                                 const int32_t* tpos, float* out_t, int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream) {
    if (num_edges < 0 || S <= 0 || Sout <= 0) return GML_E_BADARG;
    if (S != Sout || S < 2 || S > 16) return GML_E_UNSUPPORTED;
    if (num_edges == 0) return GML_OK;
    if (!ea || !w1 || !w2 || !w3 || !w4 || !out) return GML_E_BADARG;
    if (S <= 8 && (((uintptr_t)ea | (uintptr_t)out | (uintptr_t)out_t) & 15) != 0) return GML_E_BADARG;
    if (S > 8 && S % 4 == 0 && (((uintptr_t)ea | (uintptr_t)out | (uintptr_t)out_t) & 15) != 0) return GML_E_BADARG;
    if (out_t && !tpos) return GML_E_BADARG;
    // the second (source-order) copy is scattered through one buffer descriptor: 32-bit byte offsets
    if (out_t && (uint64_t)num_edges * (uint64_t)S * 4u >= 0xffffff00ull) return GML_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    switch (S) {
        case 2: return one_go<2>(ea, w1, w2, w3, w4, out, tpos, out_t, num_edges, st);
        case 3: return one_go<3>(ea, w1, w2, w3, w4, out, tpos, out_t, num_edges, st);
        case 4: return one_go<4>(ea, w1, w2, w3, w4, out, tpos, out_t, num_edges, st);
        case 5: return one_go<5>(ea, w1, w2, w3, w4, out, tpos, out_t, num_edges, st);
        case 6: return one_go<6>(ea, w1, w2, w3, w4, out, tpos, out_t, num_edges, st);
        case 7: return one_go<7>(ea, w1, w2, w3, w4, out, tpos, out_t, num_edges, st);
        case 8: return one_go<8>(ea, w1, w2, w3, w4, out, tpos, out_t, num_edges, st);
#define GML_C16X6(SV) case SV: return gml_launch_edge_chain16x6_fwd<SV>(ea, w1, w2, w3, w4, out, tpos, out_t, num_edges, st);
        GML_C16X6(9) GML_C16X6(10) GML_C16X6(11) GML_C16X6(12) GML_C16X6(13) GML_C16X6(14) GML_C16X6(15) GML_C16X6(16)
    }
    return GML_E_UNSUPPORTED;
}
