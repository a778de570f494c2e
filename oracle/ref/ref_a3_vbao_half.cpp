// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A3 (VBAO) with FEATURE_FLAG_HALF_RESOLUTION (Macros: SSAO_OPTION_HALF_RESOLUTION = 1 x SSAO_ALGORITHM = 2,
// ScreenSpaceAmbientOcclusion.cpp:473-481; SSAO_ComputeAmbientOcclusion.fx:68-75,132-236): see ref_a3_gtao_half.cpp.
#define SSAO_OPTION_HALF_RESOLUTION 1
#define SSAO_ALGORITHM 2
#define A3_NS a3_vbao_half
#define A3_ENTRY ref_ssao_compute_ao_vbao_half
#include "ref_a3_body.inc"
