// explicit instantiations of the matrix-core edge-branch kernels (S = Sout <= 8)
#include "gml_edge_chain_impl.h"
GML_DEFINE_EDGE_CHAIN(5)
GML_DEFINE_EDGE_CHAIN(6)
GML_DEFINE_EDGE_CHAIN(7)
GML_DEFINE_EDGE_CHAIN(8)
