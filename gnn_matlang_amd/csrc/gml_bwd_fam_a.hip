// explicit instantiations of the fused backward kernel (S supports, NFB/NOB = 16-wide Fin/Fout blocks)
#include "gml_spectconv_bwd_impl.h"
GML_DEFINE_BWD(8, 2, 2)
GML_DEFINE_BWD(8, 1, 2)
GML_DEFINE_BWD(4, 2, 2)
GML_DEFINE_BWD(4, 1, 2)
