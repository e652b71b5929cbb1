// explicit instantiations of the 128-row forward kernel, S = 12 (counting.py)
#include "gml_spectconv_fwd2_impl.h"
GML_DEFINE_FWD2(12, 2)
GML_DEFINE_FWD2(12, 1)
GML_DEFINE_SPMM2(12)
