// TEST INFRASTRUCTURE ONLY (oracle/_ref).  The PBR shade with ENABLE_SHADOWS = 1 and PCF_FILTER_SIZE = 5 (PBR_Renderer.cpp:1420-1434; PCFKernelSize default 3,
// PBR_Renderer.hpp:227): punctual lights with ShadowMapIndex >= 0 are attenuated by FilterShadowMapFixedPCF (Shaders/Common/public/PCF.fxh:7-152).
#define ENABLE_SHADOWS 1
#define PCF_FILTER_SIZE 5
#define pbr pbr_s5
#define ShadeAttribs ShadeAttribs_s5
#define ref_pbr_shade ref_pbr_shade_shadows5
#define ref_sizeof_pbr_light_attribs ref_sizeof_pbr_light_attribs_s5
#include "ref_p_pbr_shade.cpp"
