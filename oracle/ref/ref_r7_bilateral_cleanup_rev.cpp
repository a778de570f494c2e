// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R7 with reversed depth: the neighbour test IsReflectionSample (SSR_ComputeBilateralCleanup.fx:79) goes through IsBackground.
#define SSR_OPTION_INVERTED_DEPTH 1
#define r7 r7_rev
#define ref_ssr_bilateral_cleanup ref_ssr_bilateral_cleanup_rev
#include "ref_r7_bilateral_cleanup.cpp"
