// the HAD form of the bf16x3 fused backward (gml_bwd3_fam_h.hip) for a layer whose input needs no gradient (the model's first layer:
// Zinc12k.py:338): output stage inside, no dX projection
#include "gml_spectconv_bwd3_impl.h"

int gml_launch_bwd3_had_nodx(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st) {
    GML_ALLOW_BIG_LDS(rc, (&gml_k_spectconv_bwd3<8, 2, 8, true, false, 2, true>), 160 * 1024)
    if (rc != hipSuccess) return (int)rc;
    hipLaunchKernelGGL((gml_k_spectconv_bwd3<8, 2, 8, true, false, 2, true>), grid, dim3(64 * 8), lds, st, p);
    return gml_launch_status();
}
