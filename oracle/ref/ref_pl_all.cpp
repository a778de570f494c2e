// TEST INFRASTRUCTURE ONLY (oracle/_ref).  The layered PBR shade (ref_pl_body.inc), permutation "all": the reference's PBR_Shading.fxh compiled with
// ENABLE_CLEAR_COAT = 1, ENABLE_SHEEN = 1, ENABLE_ANISOTROPY = 1, ENABLE_IRIDESCENCE = 1, ENABLE_TRANSMISSION = 1 (PBR_Renderer.cpp:1436-1452 defines them from the PSO flags).
#define ENABLE_CLEAR_COAT 1
#define ENABLE_SHEEN 1
#define ENABLE_ANISOTROPY 1
#define ENABLE_IRIDESCENCE 1
#define ENABLE_TRANSMISSION 1
#define PL_NS pbr_layers_all
#define PL_ENTRY ref_pbr_shade_layers_all
#include "ref_pl_body.inc"
