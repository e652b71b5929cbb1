// chunked ring forward (gml_spectconv_fwd4_impl.h), S = 6: Fin <= 32 and Fin <= 48
#include "gml_spectconv_fwd4_impl.h"
GML_DEFINE_FWD4(6, 0, 2)
GML_DEFINE_FWD4(6, 1, 2)
