// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R4: SSR_ComputeIntersection.fx (ComputeIntersectionPS :281), host ScreenSpaceReflection.cpp:963-999:
// blue noise XY (:977), Hi-Z via Load, both targets cleared to 0 (:993-994), executed only under the mask (depth test LESS vs the D16 mask).
#define SSR_OPTION_PREVIOUS_FRAME 0
#define R4NS r4
#define R4FN ref_ssr_intersection
#include "ref_r4_body.inc"
