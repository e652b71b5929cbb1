// counting.py's shape class: 12 supports, Fout <= 16 (one 16-wide output block per lane group)
#include "gml_spectconv_bwd3_impl.h"
GML_DEFINE_BWD3_N1(12, 2, 8)
GML_DEFINE_BWD3_N1(12, 1, 8)
