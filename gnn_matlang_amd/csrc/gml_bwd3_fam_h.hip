// the bf16x3 fused backward with the ML3Layer output stage inside (HAD form): ZINC's shape class, S = 8, 17 .. 32 input features,
// 30 + 2 outputs (gml_spectconv_bwd_had)
#include "gml_spectconv_bwd3_impl.h"

int gml_launch_bwd3_had_nodx(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st);   // gml_bwd3_fam_i.hip

int gml_launch_bwd3_had(const GmlBwdParams& p, dim3 grid, size_t lds, hipStream_t st) {
    if (!p.xvec || p.nmix != 4 || !p.wmix || !p.hpart || !p.dw_partial || p.Fout != 30) return GML_E_UNSUPPORTED;
    if (!p.dx) return gml_launch_bwd3_had_nodx(p, grid, lds, st);
    if (!p.dxvec) return GML_E_UNSUPPORTED;
    GML_ALLOW_BIG_LDS(rc, (&gml_k_spectconv_bwd3<8, 2, 8, true, true, 2, true>), 160 * 1024)
    if (rc != hipSuccess) return (int)rc;
    hipLaunchKernelGGL((gml_k_spectconv_bwd3<8, 2, 8, true, true, 2, true>), grid, dim3(64 * 8), lds, st, p);
    return gml_launch_status();
}
