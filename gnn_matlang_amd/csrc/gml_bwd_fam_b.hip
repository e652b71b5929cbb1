// explicit instantiations of the fused backward kernel (S supports, NFB/NOB = 16-wide Fin/Fout blocks)
#include "gml_spectconv_bwd_impl.h"
GML_DEFINE_BWD(12, 2, 1)
GML_DEFINE_BWD(12, 1, 1)
GML_DEFINE_BWD(6, 3, 2)
GML_DEFINE_BWD(6, 1, 2)
