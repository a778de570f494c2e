// api_pbr.cpp -- C ABI of the PBR shading entry and of the SSR/SSAO composite.
#include "mifx_objects.h"
#include <cstring>

using namespace mifx;

extern "C" {

mifx_status mifx_pbr_shade_execute(mifx_postfx* ctx, const mifx_gbuffer* gbuffer, const mifx_camera_attribs* camera, const mifx_pbr_shade_attribs* attribs,
                                   const mifx_ibl* ibl, const float background[4], const mifx_image2d* out_radiance, const mifx_image2d* out_specular_ibl)
{
    MIFX_REQUIRE(ctx != nullptr && gbuffer != nullptr && camera != nullptr && attribs != nullptr && ibl != nullptr && out_radiance != nullptr,
                 "mifx_pbr_shade_execute: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    const Rows rows = ctx->needed_rows(int(out_radiance->height));
    MifxKernelTimer timer(ctx, "pbr_shade_kernel"); // (includes the two cube-apron launches of the call)
    return launch_pbr_shade(ctx->stream, ctx->ibl_apron, gbuffer, *camera, *attribs, ibl, background, out_radiance, out_specular_ibl, rows.b, rows.e,
                            (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH) != 0); // background = far-plane depth of the context's convention
}

mifx_status mifx_pbr_shade_execute_with_shadows(mifx_postfx* ctx, const mifx_gbuffer* gbuffer, const mifx_camera_attribs* camera, const mifx_pbr_shade_attribs* attribs,
                                                const mifx_ibl* ibl, const mifx_pbr_shadows* shadows, const float background[4], const mifx_image2d* out_radiance,
                                                const mifx_image2d* out_specular_ibl)
{
    MIFX_REQUIRE(ctx != nullptr && gbuffer != nullptr && camera != nullptr && attribs != nullptr && ibl != nullptr && shadows != nullptr && out_radiance != nullptr,
                 "mifx_pbr_shade_execute_with_shadows: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    const Rows rows = ctx->needed_rows(int(out_radiance->height));
    return launch_pbr_shade(ctx->stream, ctx->ibl_apron, gbuffer, *camera, *attribs, ibl, background, out_radiance, out_specular_ibl, rows.b, rows.e,
                            (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH) != 0, shadows);
}

mifx_status mifx_pbr_shade_execute_layers(mifx_postfx* ctx, const mifx_gbuffer* gbuffer, const mifx_pbr_layers* layers, const mifx_camera_attribs* camera,
                                          const mifx_pbr_shade_attribs* attribs, const mifx_ibl* ibl, const mifx_pbr_shadows* shadows, const float background[4],
                                          const mifx_image2d* out_radiance, const mifx_image2d* out_specular_ibl)
{
    MIFX_REQUIRE(ctx != nullptr && gbuffer != nullptr && layers != nullptr && camera != nullptr && attribs != nullptr && ibl != nullptr && out_radiance != nullptr,
                 "mifx_pbr_shade_execute_layers: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    if (layers->flags == 0u) // no layer: the default permutation (or its shadowed variant), i.e. the default kernels
        return shadows ? mifx_pbr_shade_execute_with_shadows(ctx, gbuffer, camera, attribs, ibl, shadows, background, out_radiance, out_specular_ibl)
                       : mifx_pbr_shade_execute(ctx, gbuffer, camera, attribs, ibl, background, out_radiance, out_specular_ibl);
    const Rows rows = ctx->needed_rows(int(out_radiance->height));
    MifxKernelTimer timer(ctx, "pbr_shade_layers_kernel");
    return launch_pbr_shade_layers(ctx->stream, ctx->ibl_apron, gbuffer, *layers, *camera, *attribs, ibl, background, out_radiance, out_specular_ibl, rows.b, rows.e,
                                   (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH) != 0, shadows);
}

mifx_status mifx_pbr_layers_from_material_info(const void* material_info, uint64_t bytes, uint32_t layer_flags, int32_t enable_volume, uint32_t num_texture_attribs,
                                               mifx_pbr_layers* layers, mifx_pbr_material_basic_attribs* out_basic)
{
    static_assert(sizeof(mifx_pbr_material_sheen_attribs) == 16 && sizeof(mifx_pbr_material_anisotropy_attribs) == 16 && sizeof(mifx_pbr_material_iridescence_attribs) == 16 &&
                      sizeof(mifx_pbr_material_transmission_attribs) == 16,
                  "PBR_Structures.fxh:184-223");
    MIFX_REQUIRE(material_info != nullptr && layers != nullptr, "mifx_pbr_layers_from_material_info: null argument");
    const uint32_t known = MIFX_PBR_LAYER_CLEAR_COAT | MIFX_PBR_LAYER_SHEEN | MIFX_PBR_LAYER_ANISOTROPY | MIFX_PBR_LAYER_IRIDESCENCE | MIFX_PBR_LAYER_TRANSMISSION;
    MIFX_REQUIRE((layer_flags & ~known) == 0u, "mifx_pbr_layers_from_material_info: unknown layer flags 0x%x", layer_flags & ~known);
    MIFX_REQUIRE(num_texture_attribs <= 64u, "mifx_pbr_layers_from_material_info: %u texture attribute blocks", num_texture_attribs);
    // PBRMaterialShaderInfo (PBR_Structures.fxh:291-317): the order of the optional blocks is the order of the struct
    size_t off = sizeof(mifx_pbr_material_basic_attribs);
    const size_t offSheen = off;        off += (layer_flags & MIFX_PBR_LAYER_SHEEN) ? sizeof(mifx_pbr_material_sheen_attribs) : 0;
    const size_t offAnisotropy = off;   off += (layer_flags & MIFX_PBR_LAYER_ANISOTROPY) ? sizeof(mifx_pbr_material_anisotropy_attribs) : 0;
    const size_t offIridescence = off;  off += (layer_flags & MIFX_PBR_LAYER_IRIDESCENCE) ? sizeof(mifx_pbr_material_iridescence_attribs) : 0;
    off += (layer_flags & MIFX_PBR_LAYER_TRANSMISSION) ? sizeof(mifx_pbr_material_transmission_attribs) : 0;
    off += enable_volume ? 32u : 0u;               // PBRMaterialVolumeAttribs (:228-239)
    off += size_t(num_texture_attribs) * 48u;      // PBRMaterialTextureAttribs (:244-254)
    (void)offSheen;
    MIFX_REQUIRE(bytes == off, "mifx_pbr_layers_from_material_info: PBRMaterialShaderInfo with layers 0x%x%s and %u texture blocks is %zu bytes, got %llu", layer_flags,
                 enable_volume ? " + volume" : "", num_texture_attribs, off, static_cast<unsigned long long>(bytes));
    const unsigned char* p = static_cast<const unsigned char*>(material_info);
    layers->flags = layer_flags;
    if (layer_flags & MIFX_PBR_LAYER_ANISOTROPY)
    {
        mifx_pbr_material_anisotropy_attribs a;
        std::memcpy(&a, p + offAnisotropy, sizeof(a));
        layers->anisotropy_rotation = a.Rotation;
    }
    if (layer_flags & MIFX_PBR_LAYER_IRIDESCENCE)
    {
        mifx_pbr_material_iridescence_attribs a;
        std::memcpy(&a, p + offIridescence, sizeof(a));
        layers->iridescence_ior = a.IOR;
    }
    if (out_basic != nullptr) std::memcpy(out_basic, p, sizeof(*out_basic));
    return MIFX_OK;
}

mifx_status mifx_pbr_shade_attribs_from_frame_attribs(const void* frame_attribs, uint64_t frame_attribs_bytes, uint32_t max_lights, uint32_t max_shadow_maps,
                                                      const mifx_pbr_material_basic_attribs* material, mifx_pbr_shade_attribs* out_attribs, mifx_camera_attribs* out_camera,
                                                      mifx_pbr_shadow_map_info* out_shadow_maps)
{
    static_assert(sizeof(mifx_pbr_renderer_shader_parameters) == 144 && sizeof(mifx_pbr_material_basic_attribs) == 96 && sizeof(mifx_pbr_loading_animation_parameters) == 48,
                  "PBR_Structures.fxh:111-180");
    MIFX_REQUIRE(frame_attribs != nullptr && out_attribs != nullptr && out_camera != nullptr, "mifx_pbr_shade_attribs_from_frame_attribs: null argument");
    MIFX_REQUIRE(max_lights <= MIFX_PBR_MAX_LIGHTS && max_shadow_maps <= MIFX_PBR_MAX_SHADOW_MAPS,
                 "mifx_pbr_shade_attribs_from_frame_attribs: PBR_MAX_LIGHTS %u / PBR_MAX_SHADOW_MAPS %u exceed %d / %d", max_lights, max_shadow_maps, MIFX_PBR_MAX_LIGHTS,
                 MIFX_PBR_MAX_SHADOW_MAPS);
    // RenderPBR_Structures.fxh:11-24
    const size_t offRenderer = 2 * sizeof(mifx_camera_attribs), offLights = offRenderer + sizeof(mifx_pbr_renderer_shader_parameters);
    const size_t offShadows = offLights + size_t(max_lights) * sizeof(mifx_pbr_light_attribs), total = offShadows + size_t(max_shadow_maps) * sizeof(mifx_pbr_shadow_map_info);
    MIFX_REQUIRE(frame_attribs_bytes == total, "mifx_pbr_shade_attribs_from_frame_attribs: PBRFrameAttribs with %u lights and %u shadow maps is %zu bytes, got %llu", max_lights,
                 max_shadow_maps, total, static_cast<unsigned long long>(frame_attribs_bytes));
    const unsigned char* p = static_cast<const unsigned char*>(frame_attribs);
    mifx_pbr_renderer_shader_parameters r;
    std::memcpy(out_camera, p, sizeof(mifx_camera_attribs));
    std::memcpy(&r, p + offRenderer, sizeof(r));
    MIFX_REQUIRE(r.LightCount >= 0 && uint32_t(r.LightCount) <= max_lights, "mifx_pbr_shade_attribs_from_frame_attribs: Renderer.LightCount %d, PBR_MAX_LIGHTS %u", r.LightCount, max_lights);
    MIFX_REQUIRE(r.DebugView == 0, "mifx_pbr_shade_attribs_from_frame_attribs: DebugView %d (the debug views are not part of this path)", r.DebugView);
    mifx_pbr_shade_attribs a{};
    for (int i = 0; i < 4; ++i) a.IBLScale[i] = r.IBLScale[i];
    a.OcclusionStrength      = r.OcclusionStrength;
    a.EmissionScale          = r.EmissionScale;
    a.PrefilteredCubeLastMip = r.PrefilteredCubeLastMip;
    a.LightCount             = r.LightCount;
    if (r.LightCount > 0) std::memcpy(a.Lights, p + offLights, size_t(r.LightCount) * sizeof(mifx_pbr_light_attribs));
    a.Workflow = material ? material->Workflow : MIFX_PBR_WORKFLOW_METALLIC_ROUGHNESS;
    MIFX_REQUIRE(a.Workflow == MIFX_PBR_WORKFLOW_METALLIC_ROUGHNESS || a.Workflow == MIFX_PBR_WORKFLOW_SPECULAR_GLOSSINESS,
                 "mifx_pbr_shade_attribs_from_frame_attribs: Workflow %d (PBR_WORKFLOW_UNLIT has no lighting pass)", a.Workflow);
    if (out_shadow_maps != nullptr && max_shadow_maps > 0) std::memcpy(out_shadow_maps, p + offShadows, size_t(max_shadow_maps) * sizeof(mifx_pbr_shadow_map_info));
    *out_attribs = a;
    return MIFX_OK;
}

mifx_status mifx_pbr_shade_execute_frame_attribs(mifx_postfx* ctx, const mifx_gbuffer* gbuffer, const void* frame_attribs, uint64_t frame_attribs_bytes, uint32_t max_lights,
                                                 uint32_t max_shadow_maps, const mifx_pbr_material_basic_attribs* material, const mifx_ibl* ibl,
                                                 const mifx_shadow_map_array* shadow_map, uint32_t pcf_filter_size, const float background[4], const mifx_image2d* out_radiance,
                                                 const mifx_image2d* out_specular_ibl)
{
    MIFX_REQUIRE(ctx != nullptr, "mifx_pbr_shade_execute_frame_attribs: null context");
    MIFX_REQUIRE(shadow_map != nullptr || max_shadow_maps == 0, "mifx_pbr_shade_execute_frame_attribs: %u shadow maps described but no shadow-map array bound", max_shadow_maps);
    mifx_pbr_shade_attribs   attribs;
    mifx_camera_attribs      camera;
    mifx_pbr_shadow_map_info infos[MIFX_PBR_MAX_SHADOW_MAPS];
    MIFX_CHECK(mifx_pbr_shade_attribs_from_frame_attribs(frame_attribs, frame_attribs_bytes, max_lights, max_shadow_maps, material, &attribs, &camera, infos));
    if (shadow_map == nullptr) return mifx_pbr_shade_execute(ctx, gbuffer, &camera, &attribs, ibl, background, out_radiance, out_specular_ibl);
    const mifx_pbr_shadows sh{shadow_map, infos, max_shadow_maps, pcf_filter_size};
    return mifx_pbr_shade_execute_with_shadows(ctx, gbuffer, &camera, &attribs, ibl, &sh, background, out_radiance, out_specular_ibl);
}

mifx_status mifx_pbr_shade_execute_native(mifx_postfx* ctx, const mifx_gbuffer_native* gbuffer, const mifx_camera_attribs* camera, const mifx_pbr_shade_attribs* attribs,
                                          const mifx_ibl* ibl, const float background[4], const mifx_native_image* out_radiance, const mifx_native_image* out_specular_ibl)
{
    MIFX_REQUIRE(ctx != nullptr && gbuffer != nullptr && camera != nullptr && attribs != nullptr && ibl != nullptr && out_radiance != nullptr,
                 "mifx_pbr_shade_execute_native: null argument");
    MIFX_REQUIRE(gbuffer->base_color && gbuffer->normal && gbuffer->material && gbuffer->depth, "mifx_pbr_shade_execute_native: base_color, normal, material and depth are required");
    MIFX_REQUIRE(ctx->need.empty() && ctx->band.empty(), "mifx_pbr_shade_execute_native: not available inside a row band");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_pbr_shade_native(ctx->stream, ctx->ibl_apron, gbuffer, *camera, *attribs, ibl, background, out_radiance, out_specular_ibl,
                                   (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH) != 0);
}

mifx_status mifx_pbr_specgloss_to_material(mifx_postfx* ctx, const mifx_image2d* base_color, const mifx_image2d* physical_desc, const mifx_image2d* out_material)
{
    MIFX_REQUIRE(ctx != nullptr && base_color != nullptr && physical_desc != nullptr && out_material != nullptr, "mifx_pbr_specgloss_to_material: null argument");
    Img bc, pd, out;
    MIFX_CHECK(to_img(out_material, MIFX_FORMAT_F32X4, "out_material", out));
    MIFX_CHECK(to_img_wh(base_color, MIFX_FORMAT_F32X4, out_material->width, out_material->height, "base_color", bc));
    MIFX_CHECK(to_img_wh(physical_desc, MIFX_FORMAT_F32X4, out_material->width, out_material->height, "physical_desc", pd));
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_specgloss_material(ctx->stream, bc, pd, out);
}

mifx_status mifx_composite_execute(mifx_postfx* ctx, const mifx_composite_attribs* attribs, const mifx_image2d* out)
{
    MIFX_REQUIRE(ctx != nullptr && attribs != nullptr && out != nullptr, "mifx_composite_execute: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    MifxKernelTimer timer(ctx, "composite_kernel");
    const Rows rows = ctx->needed_rows(int(out->height));
    return launch_composite(ctx->stream, *attribs, out, rows.b, rows.e);
}

mifx_status mifx_postfx_set_static_ibl(mifx_postfx* ctx, int32_t enable)
{
    MIFX_REQUIRE(ctx != nullptr, "mifx_postfx_set_static_ibl: null context");
    ctx->ibl_apron.keep  = enable != 0;
    ctx->ibl_apron.valid = false; // (re-made at the next shade either way)
    return MIFX_OK;
}

mifx_status mifx_ibl_precompute_brdf_lut(mifx_postfx* ctx, const mifx_image2d* out_lut, uint32_t num_samples)
{
    MIFX_REQUIRE(ctx != nullptr && num_samples > 0, "mifx_ibl_precompute_brdf_lut: bad argument");
    Img out;
    MIFX_CHECK(to_img(out_lut, MIFX_FORMAT_F32X2, "out_lut", out));
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_ibl_brdf_lut(ctx->stream, out, num_samples);
}

mifx_status mifx_ibl_prefilter_env_map(mifx_postfx* ctx, const mifx_cubemap* env, void* out, uint32_t out_size, float roughness, uint32_t num_samples)
{
    MIFX_REQUIRE(ctx != nullptr && env != nullptr && out != nullptr && out_size > 0 && num_samples > 0, "mifx_ibl_prefilter_env_map: bad argument");
    ctx->ibl_apron.valid = false; // (this call rewrites a map the shade may hold a working copy of)
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_ibl_prefilter(ctx->stream, env, nullptr, out, out_size, roughness, num_samples);
}

mifx_status mifx_ibl_prefilter_env_map_sphere(mifx_postfx* ctx, const mifx_spheremap* env, void* out, uint32_t out_size, float roughness, uint32_t num_samples)
{
    MIFX_REQUIRE(ctx != nullptr && env != nullptr && out != nullptr && out_size > 0 && num_samples > 0, "mifx_ibl_prefilter_env_map_sphere: bad argument");
    ctx->ibl_apron.valid = false; // (this call rewrites a map the shade may hold a working copy of)
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_ibl_prefilter(ctx->stream, nullptr, env, out, out_size, roughness, num_samples);
}

mifx_status mifx_ibl_compute_irradiance_map(mifx_postfx* ctx, const mifx_cubemap* env, void* out, uint32_t out_size, uint32_t num_samples)
{
    MIFX_REQUIRE(ctx != nullptr && env != nullptr && out != nullptr && out_size > 0 && num_samples > 0, "mifx_ibl_compute_irradiance_map: bad argument");
    ctx->ibl_apron.valid = false; // (this call rewrites a map the shade may hold a working copy of)
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_ibl_irradiance(ctx->stream, env, nullptr, out, out_size, num_samples);
}

mifx_status mifx_ibl_compute_irradiance_map_sphere(mifx_postfx* ctx, const mifx_spheremap* env, void* out, uint32_t out_size, uint32_t num_samples)
{
    MIFX_REQUIRE(ctx != nullptr && env != nullptr && out != nullptr && out_size > 0 && num_samples > 0, "mifx_ibl_compute_irradiance_map_sphere: bad argument");
    ctx->ibl_apron.valid = false; // (this call rewrites a map the shade may hold a working copy of)
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_ibl_irradiance(ctx->stream, nullptr, env, out, out_size, num_samples);
}

mifx_status mifx_envmap_render(mifx_postfx* ctx, const mifx_envmap_render_attribs* attribs, const mifx_tone_mapping_attribs* tone_mapping, const mifx_camera_attribs* camera,
                               const mifx_camera_attribs* prev_camera, const mifx_image2d* depth, const mifx_image2d* color, const mifx_image2d* motion)
{
    MIFX_REQUIRE(ctx != nullptr && attribs != nullptr && tone_mapping != nullptr && camera != nullptr && prev_camera != nullptr && depth != nullptr && color != nullptr,
                 "mifx_envmap_render: null argument");
    MIFX_REQUIRE(attribs->env_map != nullptr || attribs->sphere_map != nullptr, "mifx_envmap_render: the environment map must not be null (EnvMapRenderer.cpp:208-212)");
    MIFX_REQUIRE((attribs->options & ~7u) == 0, "mifx_envmap_render: unknown option flags 0x%x", attribs->options);
    MIFX_REQUIRE(tone_mapping->iToneMappingMode >= 0 && tone_mapping->iToneMappingMode <= MIFX_TONE_MAPPING_MODE_COMMERCE, "mifx_envmap_render: unknown tone mapping mode %d",
                 tone_mapping->iToneMappingMode);
    Img d, c, m{nullptr, 0, 0, 0, 0, 0};
    MIFX_CHECK(to_img(color, MIFX_FORMAT_F32X4, "color", c));
    MIFX_CHECK(to_img_wh(depth, MIFX_FORMAT_F32, color->width, color->height, "depth", d));
    if (motion != nullptr) MIFX_CHECK(to_img_wh(motion, MIFX_FORMAT_F32X2, color->width, color->height, "motion", m));
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_envmap(ctx->stream, *attribs, *tone_mapping, *camera, *prev_camera, d, c, m);
}

uint32_t mifx_native_format_texel_size(uint32_t format) { return native_texel_size(format); }

mifx_status mifx_image_import(mifx_postfx* ctx, const mifx_native_image* src, const mifx_image2d* dst)
{
    MIFX_REQUIRE(ctx != nullptr && src != nullptr && dst != nullptr, "mifx_image_import: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_image_import(ctx->stream, src, dst);
}
mifx_status mifx_image_export(mifx_postfx* ctx, const mifx_image2d* src, const mifx_native_image* dst)
{
    MIFX_REQUIRE(ctx != nullptr && src != nullptr && dst != nullptr, "mifx_image_export: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_image_export(ctx->stream, src, dst);
}

} // extern "C"
