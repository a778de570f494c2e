// TEST INFRASTRUCTURE ONLY (oracle/_ref).  E1 with an equirectangular environment map (ENV_MAP_TYPE_SPHERE, EnvMapRenderer.cpp:214-216; EnvMap.psh:33-35), Hydrogent's
// option set (tone mapping NONE, motion vectors).
#define ENV_MAP_TYPE 1
#define TONE_MAPPING_MODE 0
#define CONVERT_OUTPUT_TO_SRGB 0
#define COMPUTE_MOTION_VECTORS 1
#define E1NS e1s
#define E1FN ref_envmap_sphere
#include "ref_e1_body.inc"
