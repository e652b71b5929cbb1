// explicit instantiations of the ML3Layer edge-branch kernels (S = Sout)
#include "gml_edge_mlp_impl.h"
GML_DEFINE_EDGE_MLP(9)
GML_DEFINE_EDGE_MLP(10)
GML_DEFINE_EDGE_MLP(11)
