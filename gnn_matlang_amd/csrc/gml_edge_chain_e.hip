// Edge branches of a stack of ML3Layers in one pass over the raw supports (gml_edge_mlp_fwd_stack): S = Sout in {4, 8}, 2..4 layers
#include "gml_edge_chain_impl.h"
#include <stdlib.h>

template <int S, int L>
static int stack_go(const uint32_t* es, const float* const* w1, const float* const* w2, const float* const* w3,
                    const float* const* w4, float* const* out, int64_t E, hipStream_t st) {
    GmlChainStack<L> a;
    for (int l = 0; l < L; ++l) { a.w1[l] = w1[l]; a.w2[l] = w2[l]; a.w3[l] = w3[l]; a.w4[l] = w4[l]; a.out[l] = out[l]; }
    return gml_launch_edge_chain_fwd_stack<S, L>(es, a, E, st);
}

// out[l] = relu(W4_l [relu(W1_l e) ; tanh(W2_l e) * tanh(W3_l e)]) for l < nlayers, e = the rows of ea whose bf16 (hi, lo) split is
// ea_split (gml_edge_presplit).  The pointer arrays live on the HOST.  GML_E_UNSUPPORTED outside (S = Sout in {4, 8},
// 2 <= nlayers <= 4): the caller launches gml_edge_mlp_fwd per layer.
extern "C" int gml_edge_mlp_fwd_stack(const void* ea_split, int32_t nlayers, const float* const* w1, const float* const* w2,
                                      const float* const* w3, const float* const* w4, float* const* out,
                                      int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream) {
    if (num_edges < 0 || S <= 0 || Sout <= 0 || nlayers <= 0 || !w1 || !w2 || !w3 || !w4 || !out) return GML_E_BADARG;
    static const bool valu = [] { const char* e = getenv("GML_EDGE_VALU"); return e && e[0] == '1'; }();   /* (A/B: the fp32 VALU kernels) */
    if (valu || S != Sout || (S != 8 && S != 4) || nlayers < 2 || nlayers > 4) return GML_E_UNSUPPORTED;
    if (num_edges == 0) return GML_OK;
    if (!ea_split || (((uintptr_t)ea_split) & 15) != 0) return GML_E_BADARG;
    for (int l = 0; l < nlayers; ++l)
        if (!w1[l] || !w2[l] || !w3[l] || !w4[l] || !out[l] || (((uintptr_t)out[l]) & 15) != 0) return GML_E_BADARG;
    const uint32_t* es = (const uint32_t*)ea_split;
    hipStream_t st = (hipStream_t)stream;
#define GML_STACK_GO(SV, LV) if (S == SV && nlayers == LV) return stack_go<SV, LV>(es, w1, w2, w3, w4, out, num_edges, st);
    GML_STACK_GO(8, 2) GML_STACK_GO(8, 3) GML_STACK_GO(8, 4) GML_STACK_GO(4, 2) GML_STACK_GO(4, 3) GML_STACK_GO(4, 4)
    return GML_E_UNSUPPORTED;
}
