// explicit instantiations of the fused backward with the LDS-DMA landing ring
#include "gml_spectconv_bwd4_impl.h"
GML_DEFINE_BWD4(8, 1)
GML_DEFINE_BWD4(4, 2)
GML_DEFINE_BWD4(4, 1)
