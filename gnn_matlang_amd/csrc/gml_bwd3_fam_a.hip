// explicit instantiations of the bf16x3 fused backward kernel, third layout (8 waves / 128-row groups)
#include "gml_spectconv_bwd3_impl.h"
GML_DEFINE_BWD3(8, 2, 8)
GML_DEFINE_BWD3(8, 1, 8)
