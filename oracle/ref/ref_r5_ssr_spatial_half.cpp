// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R5 with FEATURE_FLAG_HALF_RESOLUTION: the Poisson taps address the half-resolution ray textures
// (SSR_ComputeSpatialReconstruction.fx:153-154); the pass itself stays at full resolution under the full-resolution mask (ScreenSpaceReflection.cpp:1026).
#define SSR_OPTION_HALF_RESOLUTION 1
#define r5 r5_half
#define ref_ssr_spatial_reconstruction ref_ssr_spatial_reconstruction_half
#include "ref_r5_ssr_spatial.cpp"
