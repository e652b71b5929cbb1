// 33 .. 48 input features in ONE launch (NFB = 3: a second 32-feature image of W and of the X rows, round 5):
// sr25.py:252-262 (S = 6, 32 + 16 wide), mutag.py:272-288 (S = 4, 24 + 24 wide)
#include "gml_spectconv_bwd3_impl.h"
GML_DEFINE_BWD3(6, 3, 8)
GML_DEFINE_BWD3(4, 3, 8)
