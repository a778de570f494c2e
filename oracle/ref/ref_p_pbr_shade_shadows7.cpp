// TEST INFRASTRUCTURE ONLY (oracle/_ref).  The PBR shade with ENABLE_SHADOWS = 1 and PCF_FILTER_SIZE = 7 (PBR_Renderer.cpp:1420-1434):
// punctual lights with ShadowMapIndex >= 0 are attenuated by FilterShadowMapFixedPCF (Shaders/Common/public/PCF.fxh:7-152).
#define ENABLE_SHADOWS 1
#define PCF_FILTER_SIZE 7
#define pbr pbr_s7
#define ShadeAttribs ShadeAttribs_s7
#define ref_pbr_shade ref_pbr_shade_shadows7
#define ref_sizeof_pbr_light_attribs ref_sizeof_pbr_light_attribs_s7
#include "ref_p_pbr_shade.cpp"
