// TEST INFRASTRUCTURE ONLY (oracle/_ref).  C3 with FEATURE_FLAG_REVERSED_DEPTH (Macros: POSTFX_OPTION_INVERTED_DEPTH = 1): the closest depth is the largest one, ComputeClosestMotion.fx:5-9,36-40.
#define POSTFX_OPTION_INVERTED_DEPTH 1
#define c3 c3_rev
#define ref_closest_motion ref_closest_motion_rev
#include "ref_c3_closest_motion.cpp"
