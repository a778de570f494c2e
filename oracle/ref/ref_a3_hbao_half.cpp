// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A3 (HBAO) with FEATURE_FLAG_HALF_RESOLUTION (Macros: SSAO_OPTION_HALF_RESOLUTION = 1 x SSAO_ALGORITHM = 1,
// ScreenSpaceAmbientOcclusion.cpp:473-481; SSAO_ComputeAmbientOcclusion.fx:68-75,132-236): see ref_a3_gtao_half.cpp.
#define SSAO_OPTION_HALF_RESOLUTION 1
#define SSAO_ALGORITHM 1
#define A3_NS a3_hbao_half
#define A3_ENTRY ref_ssao_compute_ao_hbao_half
#include "ref_a3_body.inc"
