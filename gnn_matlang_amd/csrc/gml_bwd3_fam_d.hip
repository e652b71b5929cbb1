// 4 waves / 64-row groups: two workgroups per CU
#include "gml_spectconv_bwd3_impl.h"
GML_DEFINE_BWD3(8, 2, 4)
GML_DEFINE_BWD3(8, 1, 4)
