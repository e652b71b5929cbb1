// explicit instantiations of the 128-row forward kernel
#include "gml_spectconv_fwd2_impl.h"
GML_DEFINE_FWD2(8, 2)
GML_DEFINE_FWD2(8, 1)
GML_DEFINE_FWD2(4, 2)
GML_DEFINE_FWD2(4, 1)
GML_DEFINE_SPMM2(8)
GML_DEFINE_SPMM2(4)
