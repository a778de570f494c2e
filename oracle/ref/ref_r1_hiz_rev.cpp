// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R1 with reversed depth (SSR_OPTION_INVERTED_DEPTH = 1, ScreenSpaceReflection.cpp:73,473): ClosestDepth = max, DepthFarPlane = 0 (SSR_Common.fxh:6-12).
#define SSR_OPTION_INVERTED_DEPTH 1
#define r1 r1_rev
#define ref_ssr_hiz_mip ref_ssr_hiz_mip_rev
#include "ref_r1_hiz.cpp"
