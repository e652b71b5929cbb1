// explicit instantiations of the matrix-core edge-branch kernels (S = Sout <= 8)
#include "gml_edge_chain_impl.h"
GML_DEFINE_EDGE_CHAIN(1)
GML_DEFINE_EDGE_CHAIN(2)
GML_DEFINE_EDGE_CHAIN(3)
GML_DEFINE_EDGE_CHAIN(4)
