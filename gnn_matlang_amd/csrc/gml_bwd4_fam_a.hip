// explicit instantiations of the fused backward with the LDS-DMA landing ring
#include "gml_spectconv_bwd4_impl.h"
GML_DEFINE_BWD4(8, 2)
