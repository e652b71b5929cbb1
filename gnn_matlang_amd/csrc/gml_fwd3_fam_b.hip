// explicit instantiations of the 128-row forward kernel with the LDS-DMA landing ring, S = 4
#include "gml_spectconv_fwd3_impl.h"
GML_DEFINE_FWD3(4, 2)
GML_DEFINE_FWD3(4, 1)
