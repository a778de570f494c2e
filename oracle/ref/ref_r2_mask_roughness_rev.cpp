// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R2 with reversed depth: IsBackground = Depth < 1e-6 (SSR_Common.fxh:48-55).
#define SSR_OPTION_INVERTED_DEPTH 1
#define r2 r2_rev
#define ref_ssr_mask_roughness ref_ssr_mask_roughness_rev
#define ref_sizeof_ssr_attribs ref_sizeof_ssr_attribs_rev
#include "ref_r2_mask_roughness.cpp"
