// chunked ring forward (gml_spectconv_fwd4_impl.h), S = 8: Fin <= 32 (groups beyond fwd3's staging, opt-in)
#include "gml_spectconv_fwd4_impl.h"
GML_DEFINE_FWD4(8, 0, 2)
