// TEST INFRASTRUCTURE ONLY (oracle/_ref).  The layered PBR shade (ref_pl_body.inc), permutation "anisotropy": the reference's PBR_Shading.fxh compiled with
// ENABLE_CLEAR_COAT = 0, ENABLE_SHEEN = 0, ENABLE_ANISOTROPY = 1, ENABLE_IRIDESCENCE = 0, ENABLE_TRANSMISSION = 0 (PBR_Renderer.cpp:1436-1452 defines them from the PSO flags).
#define ENABLE_CLEAR_COAT 0
#define ENABLE_SHEEN 0
#define ENABLE_ANISOTROPY 1
#define ENABLE_IRIDESCENCE 0
#define ENABLE_TRANSMISSION 0
#define PL_NS pbr_layers_anisotropy
#define PL_ENTRY ref_pbr_shade_layers_anisotropy
#include "ref_pl_body.inc"
