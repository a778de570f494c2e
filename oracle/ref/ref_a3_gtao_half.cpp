// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A3 (GTAO) with FEATURE_FLAG_HALF_RESOLUTION (Macros: SSAO_OPTION_HALF_RESOLUTION = 1, ScreenSpaceAmbientOcclusion.cpp:473):
// the target and the prefiltered depth pyramid are (W / 2) x (H / 2) (:109-110, 273-274), GetInvViewportSize() = 2 / viewport (SSAO_ComputeAmbientOcclusion.fx:68-75).
#define SSAO_OPTION_HALF_RESOLUTION 1
#define SSAO_ALGORITHM 0
#define A3_NS a3_gtao_half
#define A3_ENTRY ref_ssao_compute_ao_gtao_half
#include "ref_a3_body.inc"
