// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R4 with FEATURE_FLAG_HALF_RESOLUTION (Macros: SSR_OPTION_HALF_RESOLUTION = 1, ScreenSpaceReflection.cpp:475): targets and mask
// are (W / 2) x (H / 2) (:201-213, 181-190); every half-resolution texel traces the ray of one full-resolution pixel of its 2x2 block, chosen by
// ComputeHalfResolutionOffset (SSR_ComputeIntersection.fx:283-288, PostFX_Common.fxh:45-55).
#define SSR_OPTION_HALF_RESOLUTION 1
#define SSR_OPTION_PREVIOUS_FRAME 0
#define R4NS r4_half
#define R4FN ref_ssr_intersection_half
#include "ref_r4_body.inc"
