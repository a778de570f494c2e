// api_stubs.cpp -- entry points declared in include/mifx.h whose implementation has not landed yet.
// Each returns MIFX_ERR_NOT_IMPLEMENTED loudly (never a silent fallback); the file shrinks as passes land.
#include "mifx_objects.h"

using namespace mifx;

#define MIFX_STUB(name)                                  \
    do                                                   \
    {                                                    \
        set_error(name ": not implemented in this build"); \
        return MIFX_ERR_NOT_IMPLEMENTED;                 \
    } while (0)

mifx_bloom::~mifx_bloom()
{
    for (auto* p : down) delete p;
    for (auto* p : up) delete p;
}
mifx_chain::~mifx_chain() {}

extern "C" {
mifx_status mifx_ssr_create(mifx_postfx*, mifx_ssr**) { MIFX_STUB("mifx_ssr_create"); }
void        mifx_ssr_destroy(mifx_ssr*) {}
mifx_status mifx_ssr_prepare(mifx_ssr*, mifx_postfx*, uint32_t) { MIFX_STUB("mifx_ssr_prepare"); }
mifx_status mifx_ssr_execute(mifx_ssr*, const mifx_ssr_render_attribs*) { MIFX_STUB("mifx_ssr_execute"); }
mifx_status mifx_ssr_get_output(mifx_ssr*, mifx_image2d*) { MIFX_STUB("mifx_ssr_get_output"); }
mifx_status mifx_ssr_reset_history(mifx_ssr*) { MIFX_STUB("mifx_ssr_reset_history"); }
mifx_status mifx_ssr_get_intermediate(mifx_ssr*, const char*, mifx_image2d*) { MIFX_STUB("mifx_ssr_get_intermediate"); }

mifx_status mifx_taa_create(mifx_postfx*, mifx_taa**) { MIFX_STUB("mifx_taa_create"); }
void        mifx_taa_destroy(mifx_taa*) {}
mifx_status mifx_taa_prepare(mifx_taa*, mifx_postfx*, uint32_t) { MIFX_STUB("mifx_taa_prepare"); }
mifx_status mifx_taa_execute(mifx_taa*, const mifx_taa_render_attribs*) { MIFX_STUB("mifx_taa_execute"); }
mifx_status mifx_taa_get_output(mifx_taa*, int32_t, mifx_image2d*) { MIFX_STUB("mifx_taa_get_output"); }
mifx_status mifx_taa_reset_history(mifx_taa*) { MIFX_STUB("mifx_taa_reset_history"); }
mifx_status mifx_taa_get_jitter_offset(uint32_t, uint32_t, uint32_t, float*) { MIFX_STUB("mifx_taa_get_jitter_offset"); }
mifx_status mifx_taa_get_jittered_proj_matrix(const float*, const float*, float*) { MIFX_STUB("mifx_taa_get_jittered_proj_matrix"); }

mifx_status mifx_bloom_create(mifx_postfx*, mifx_bloom**) { MIFX_STUB("mifx_bloom_create"); }
void        mifx_bloom_destroy(mifx_bloom*) {}
mifx_status mifx_bloom_prepare(mifx_bloom*, mifx_postfx*, uint32_t) { MIFX_STUB("mifx_bloom_prepare"); }
mifx_status mifx_bloom_execute(mifx_bloom*, const mifx_bloom_render_attribs*) { MIFX_STUB("mifx_bloom_execute"); }
mifx_status mifx_bloom_get_output(mifx_bloom*, mifx_image2d*) { MIFX_STUB("mifx_bloom_get_output"); }
mifx_status mifx_bloom_get_intermediate(mifx_bloom*, const char*, mifx_image2d*) { MIFX_STUB("mifx_bloom_get_intermediate"); }

mifx_status mifx_pbr_shade_execute(mifx_postfx*, const mifx_gbuffer*, const mifx_camera_attribs*, const mifx_pbr_shade_attribs*, const mifx_ibl*, const float*,
                                   const mifx_image2d*, const mifx_image2d*)
{
    MIFX_STUB("mifx_pbr_shade_execute");
}
mifx_status mifx_composite_execute(mifx_postfx*, const mifx_composite_attribs*, const mifx_image2d*) { MIFX_STUB("mifx_composite_execute"); }

mifx_status mifx_chain_create(const mifx_device_desc*, const mifx_postfx_create_info*, mifx_chain**) { MIFX_STUB("mifx_chain_create"); }
void        mifx_chain_destroy(mifx_chain*) {}
mifx_status mifx_chain_execute(mifx_chain*, const mifx_chain_frame*, const mifx_image2d*) { MIFX_STUB("mifx_chain_execute"); }
mifx_status mifx_chain_get_postfx(mifx_chain*, mifx_postfx**) { MIFX_STUB("mifx_chain_get_postfx"); }
mifx_status mifx_chain_reset_history(mifx_chain*) { MIFX_STUB("mifx_chain_reset_history"); }
}
