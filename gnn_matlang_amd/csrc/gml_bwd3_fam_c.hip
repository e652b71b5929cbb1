#include "gml_spectconv_bwd3_impl.h"
GML_DEFINE_BWD3(2, 2, 8)
GML_DEFINE_BWD3(2, 1, 8)
