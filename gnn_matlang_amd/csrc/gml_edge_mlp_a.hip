// explicit instantiations of the ML3Layer edge-branch kernels (S = Sout)
#include "gml_edge_mlp_impl.h"
GML_DEFINE_EDGE_MLP(1)
GML_DEFINE_EDGE_MLP(2)
GML_DEFINE_EDGE_MLP(3)
GML_DEFINE_EDGE_MLP(4)
GML_DEFINE_EDGE_MLP(5)
