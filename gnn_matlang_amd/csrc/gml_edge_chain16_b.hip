// explicit instantiations of the matrix-core edge-branch kernels for 8 < S = Sout <= 16
#include "gml_edge_chain16_impl.h"
GML_DEFINE_EDGE_CHAIN16(11)
GML_DEFINE_EDGE_CHAIN16(12)
