// TEST INFRASTRUCTURE ONLY (oracle/_ref).  The layered PBR shade (ref_pl_body.inc), permutation "all_shadows3": ENABLE_SHADOWS = 1 with PCF_FILTER_SIZE = 3 on top of
// ENABLE_CLEAR_COAT = 1, ENABLE_SHEEN = 1, ENABLE_ANISOTROPY = 1, ENABLE_IRIDESCENCE = 1, ENABLE_TRANSMISSION = 1.
#define ENABLE_SHADOWS 1
#define PCF_FILTER_SIZE 3
#define ENABLE_CLEAR_COAT 1
#define ENABLE_SHEEN 1
#define ENABLE_ANISOTROPY 1
#define ENABLE_IRIDESCENCE 1
#define ENABLE_TRANSMISSION 1
#define PL_NS pbr_layers_all_shadows3
#define PL_ENTRY ref_pbr_shade_layers_all_shadows3
#include "ref_pl_body.inc"
