// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A3 (GTAO) with FEATURE_FLAG_HALF_PRECISION_DEPTH (Macros: SSAO_OPTION_HALF_PRECISION_DEPTH = 1, ScreenSpaceAmbientOcclusion.cpp:474):
// the self-occlusion offset grows from 1e-5 to 5e-3 (SSAO_ComputeAmbientOcclusion.fx:145-150); the R16_UNORM storage of the depth pyramids (:96-97) is a
// storage format and, like every other intermediate format, not emulated (all planes are fp32 here).
#define SSAO_OPTION_HALF_PRECISION_DEPTH 1
#define SSAO_ALGORITHM 0
#define A3_NS a3_gtao_halfprec
#define A3_ENTRY ref_ssao_compute_ao_gtao_halfprec
#include "ref_a3_body.inc"
