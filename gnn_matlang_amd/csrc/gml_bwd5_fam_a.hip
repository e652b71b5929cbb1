// explicit instantiations of the 12-wave bf16x3 fused backward (8 compute waves with two edge passes + 4 helper waves)
#include "gml_spectconv_bwd5_impl.h"
GML_DEFINE_BWD5(2)
