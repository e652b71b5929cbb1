// gml_edge_mlp_fwd_stack6: the edge branches of a stack of ML3Layers in one pass over the raw supports, three-piece products
// (gml_edge_chain6_impl.h): S = Sout in {4, 8}, 1 .. 4 layers
#include "gml_edge_chain6_impl.h"

template <int S, int L>
static int stack6_go(const float* ea, const float* const* w1, const float* const* w2, const float* const* w3, const float* const* w4,
                     float* const* out, int64_t E, hipStream_t st) {
    GmlChain6Stack<L> a;
    for (int l = 0; l < L; ++l) { a.w1[l] = w1[l]; a.w2[l] = w2[l]; a.w3[l] = w3[l]; a.w4[l] = w4[l]; a.out[l] = out[l]; }
    return gml_launch_edge_chain6_fwd<S, L>(ea, a, nullptr, nullptr, E, st);
}

extern "C" int gml_edge_mlp_fwd_stack6(const float* ea, int32_t nlayers, const float* const* w1, const float* const* w2,
                                       const float* const* w3, const float* const* w4, float* const* out,
                                       int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream) {
    if (num_edges < 0 || S <= 0 || Sout <= 0 || nlayers <= 0 || !w1 || !w2 || !w3 || !w4 || !out) return GML_E_BADARG;
    if (S != Sout || (S != 8 && S != 4) || nlayers > 4) return GML_E_UNSUPPORTED;
    if (num_edges == 0) return GML_OK;
    if (!ea || (((uintptr_t)ea) & 15) != 0) return GML_E_BADARG;
    for (int l = 0; l < nlayers; ++l)
        if (!w1[l] || !w2[l] || !w3[l] || !w4[l] || !out[l] || (((uintptr_t)out[l]) & 15) != 0) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
#define GML_STACK6_GO(SV, LV) if (S == SV && nlayers == LV) return stack6_go<SV, LV>(ea, w1, w2, w3, w4, out, num_edges, st);
    GML_STACK6_GO(8, 1) GML_STACK6_GO(8, 2) GML_STACK6_GO(8, 3) GML_STACK6_GO(8, 4)
    GML_STACK6_GO(4, 1) GML_STACK6_GO(4, 2) GML_STACK6_GO(4, 3) GML_STACK6_GO(4, 4)
    return GML_E_UNSUPPORTED;
}
