// explicit instantiations of the fused backward kernel (S supports, NFB/NOB = 16-wide Fin/Fout blocks)
#include "gml_spectconv_bwd_impl.h"
GML_DEFINE_BWD(4, 3, 2)
GML_DEFINE_BWD(6, 2, 2)
GML_DEFINE_BWD(8, 2, 1)
GML_DEFINE_BWD(4, 4, 2)
