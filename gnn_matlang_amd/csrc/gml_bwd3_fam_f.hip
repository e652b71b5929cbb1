// S = 8 with one 16-wide output block per lane group (NOB = 1): the column-split A/B of the ZINC backward (VERDICT r03 item 2;
// profiles/r04_bwd_colsplit_ab.txt) and 8-support layers with Fout <= 16
#include "gml_spectconv_bwd3_impl.h"
GML_DEFINE_BWD3_N1(8, 2, 8)
GML_DEFINE_BWD3_N1(8, 1, 8)
