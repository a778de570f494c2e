// TEST INFRASTRUCTURE ONLY (oracle/_ref).  The layered PBR shade (ref_pl_body.inc), permutation "sheen_shadows5": ENABLE_SHADOWS = 1 with PCF_FILTER_SIZE = 5 on top of
// ENABLE_CLEAR_COAT = 0, ENABLE_SHEEN = 1, ENABLE_ANISOTROPY = 0, ENABLE_IRIDESCENCE = 0, ENABLE_TRANSMISSION = 0.
#define ENABLE_SHADOWS 1
#define PCF_FILTER_SIZE 5
#define ENABLE_CLEAR_COAT 0
#define ENABLE_SHEEN 1
#define ENABLE_ANISOTROPY 0
#define ENABLE_IRIDESCENCE 0
#define ENABLE_TRANSMISSION 0
#define PL_NS pbr_layers_sheen_shadows5
#define PL_ENTRY ref_pbr_shade_layers_sheen_shadows5
#include "ref_pl_body.inc"
