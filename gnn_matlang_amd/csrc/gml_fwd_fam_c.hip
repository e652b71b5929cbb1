// explicit instantiations of the fused forward kernel families (SC supports x FPL features/lane)
#include "gml_spectconv_impl.h"
GML_DEFINE_FWD_FAMILY(8, 8)
GML_DEFINE_FWD_FAMILY(1, 4)
