// explicit instantiations of the ML3Layer edge-branch kernels (S = Sout)
#include "gml_edge_mlp_impl.h"
GML_DEFINE_EDGE_MLP(6)
GML_DEFINE_EDGE_MLP(7)
GML_DEFINE_EDGE_MLP(8)
