// explicit instantiations of the bf16x3 fused backward kernel (S supports, NFB = 16-wide Fin blocks; Fout in (16, 32])
#include "gml_spectconv_bwd2_impl.h"
GML_DEFINE_BWD2(4, 2)
GML_DEFINE_BWD2(4, 1)
GML_DEFINE_BWD2(2, 2)
GML_DEFINE_BWD2(2, 1)
