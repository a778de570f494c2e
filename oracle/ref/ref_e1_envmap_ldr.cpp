// TEST INFRASTRUCTURE ONLY (oracle/_ref).  E1, stand-alone LDR variant: TONE_MAPPING_MODE_UNCHARTED2 + OPTION_FLAG_CONVERT_OUTPUT_TO_SRGB
// (the pow(1 / 2.2) of EnvMap.psh:57-59), motion vectors on.
#define TONE_MAPPING_MODE 4
#define CONVERT_OUTPUT_TO_SRGB 1
#define COMPUTE_MOTION_VECTORS 1
#define E1NS e1l
#define E1FN ref_envmap_ldr
#include "ref_e1_body.inc"
