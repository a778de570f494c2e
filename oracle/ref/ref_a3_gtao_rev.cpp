// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A3 (GTAO) with reversed depth (SSAO_OPTION_INVERTED_DEPTH = 1, ScreenSpaceAmbientOcclusion.cpp:72): IsBackground = Depth < 1e-6 (SSAO_Common.fxh:16-23).
#define SSAO_OPTION_INVERTED_DEPTH 1
#define SSAO_ALGORITHM 0
#define A3_NS a3_gtao_rev
#define A3_ENTRY ref_ssao_compute_ao_gtao_rev
#include "ref_a3_body.inc"
