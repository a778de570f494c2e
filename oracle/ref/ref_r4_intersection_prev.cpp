// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R4 with FEATURE_FLAG_PREVIOUS_FRAME (Macros: SSR_OPTION_PREVIOUS_FRAME = 1, ScreenSpaceReflection.cpp:474):
// the colour buffer is last frame's, so the hit is moved back along its motion vector before the radiance is read (SSR_ComputeIntersection.fx:310-314)
// and the edge vignette is taken at both positions (:230-231).
#define SSR_OPTION_PREVIOUS_FRAME 1
#define R4NS r4p
#define R4FN ref_ssr_intersection_prev
#include "ref_r4_body.inc"
