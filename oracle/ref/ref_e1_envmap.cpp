// TEST INFRASTRUCTURE ONLY (oracle/_ref).  E1: the environment-map background as Hydrogent draws it (HnRenderEnvMapTask.cpp:165-219):
// TONE_MAPPING_MODE_NONE (tone mapping happens in the post-process), no gamma, motion vectors on.
#define TONE_MAPPING_MODE 0
#define CONVERT_OUTPUT_TO_SRGB 0
#define COMPUTE_MOTION_VECTORS 1
#define E1NS e1
#define E1FN ref_envmap
#include "ref_e1_body.inc"
