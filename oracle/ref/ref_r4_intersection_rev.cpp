// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R4 with reversed depth: the ray leaves through the z plane when Direction.z < 0 and is above the surface when SurfaceDepth < Position.z (SSR_ComputeIntersection.fx:109-123).
#define SSR_OPTION_INVERTED_DEPTH 1
#define SSR_OPTION_PREVIOUS_FRAME 0
#define R4NS r4_rev
#define R4FN ref_ssr_intersection_rev
#include "ref_r4_body.inc"
