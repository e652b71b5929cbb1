#include "gml_spectconv_bwd3_impl.h"
GML_DEFINE_BWD3(6, 2, 8)
GML_DEFINE_BWD3(6, 1, 8)
GML_DEFINE_BWD3(4, 2, 8)
GML_DEFINE_BWD3(4, 1, 8)
