// ibl.hip -- (I1-I3) IBL precompute: the inputs of the split-sum lighting in pbr.hip.
//   I1 preintegrated GGX BRDF LUT   Shaders/PBR/private/PrecomputeBRDF.psh:8-50        (PBR_Renderer::PrecomputeBRDF, PBR_Renderer.cpp:548-622)
//   I2 prefiltered environment map  Shaders/PBR/private/PrefilterEnvMap.psh:40-109     (PBR_Renderer::PrecomputeCubemaps, PBR_Renderer.cpp:729-972)
//   I3 irradiance map               Shaders/PBR/private/ComputeIrradianceMap.psh:43-93
// Common sampling helpers: Shaders/PBR/private/PBR_PrecomputeCommon.fxh:10-54.  One thread per output texel, Monte-Carlo loop inside
// (these are one-time precomputes, ALU/latency bound on cache-resident cube maps).
#include "mifx_host.h"
#include "mifx_tonemap.h"
#include "mifx_pbr.h"

namespace mifx
{
MIFX_D v2 hammersley2d(unsigned i, unsigned n) // PBR_PrecomputeCommon.fxh:10-16
{
    const unsigned bits = __brev(i);
    const float    rdi  = float(bits) * 2.3283064365386963e-10f;
    return v2{float(i) / float(n), rdi};
}
MIFX_D v3 importance_sample_ggx(v2 xi, float perceptualRoughness, v3 N) // :19-37
{
    const float alpha = perceptualRoughness * perceptualRoughness;
    const float a2    = alpha * alpha;
    const float phi   = 2.0f * MIFX_PI * xi.x;
    const float cosT  = sqrtf(saturate((1.0f - xi.y) / (1.0f + (a2 - 1.0f) * xi.y)));
    const float sinT  = sqrtf(saturate(1.0f - cosT * cosT));
    const v3 H{sinT * m_cos(phi), sinT * m_sin(phi), cosT};
    const v3 up = fabsf(N.z) < 0.999f ? v3{0.0f, 0.0f, 1.0f} : v3{1.0f, 0.0f, 0.0f};
    const v3 tx = normalize(cross(up, N));
    const v3 ty = cross(N, tx);
    return tx * H.x + ty * H.y + N * H.z;
}
MIFX_D float cube_pixel_solid_angle(float w, float h) { return 4.0f * MIFX_PI / (6.0f * w * h); } // :39-42

// The environment as the three passes below see it: a cube map (ENV_MAP_TYPE_CUBE) or an equirectangular Texture2D (ENV_MAP_TYPE_SPHERE), both with a mip chain and
// the linear-clamp, mip-linear sampler of the reference (PBR_Renderer.cpp:851,906; EnvMapRenderer.cpp:176).
struct EnvK
{
    CubeK     cube;
    const v4* smip[12]; // sphere map levels, tightly packed
    int       sw, sh, smips;
    int       sphere;
};
MIFX_D v4 sphere_level_sample(const v4* im, int w, int h, float u, float v) // one level, linear clamp
{
    const Bilinear b = bilinear_uc(u * float(w), v * float(h), w, h);
    const v4 t00 = im[size_t(b.y0) * w + b.x0], t10 = im[size_t(b.y0) * w + b.x1], t01 = im[size_t(b.y1) * w + b.x0], t11 = im[size_t(b.y1) * w + b.x1];
    return t00 * b.w00 + t10 * b.w10 + t01 * b.w01 + t11 * b.w11;
}
MIFX_D v4 env_sample(const EnvK& e, v3 dir, float lod)
{
    if (!e.sphere) return cube_sample(e.cube, dir, lod);
    // TransformDirectionToSphereMapUV (ShaderUtilities.fxh:98-102)
    const float oneOverPi = 0.3183098862f;
    const float u = oneOverPi * (0.5f * atan2f(dir.z, dir.x)) + 0.5f, v = oneOverPi * asinf(dir.y) + 0.5f;
    lod = clampf(lod, 0.0f, float(e.smips - 1));
    const int   l0 = int(floorf(lod)), l1 = l0 + 1 < e.smips ? l0 + 1 : l0;
    const float f  = lod - float(l0);
    const v4 c0 = sphere_level_sample(e.smip[l0], max(e.sw >> l0, 1), max(e.sh >> l0, 1), u, v);
    if (f == 0.0f || l1 == l0) return c0;
    const v4 c1 = sphere_level_sample(e.smip[l1], max(e.sw >> l1, 1), max(e.sh >> l1, 1), u, v);
    return c0 + (c1 - c0) * f;
}
// solid angle of one texel of the environment (PBR_PrecomputeCommon.fxh:38-48); gamma: 1 in PrefilterEnvMap.psh:83, 0.5 in ComputeIrradianceMap.psh:71
MIFX_D float env_pixel_solid_angle(const EnvK& e, v3 L, float gamma)
{
    if (!e.sphere) return cube_pixel_solid_angle(float(e.cube.size), float(e.cube.size));
    const float theta = acosf(L.y), dTheta = MIFX_PI / float(e.sw), dPhi = 2.0f * MIFX_PI / float(e.sh);
    return dPhi * (cosf(theta - 0.5f * dTheta * gamma) - cosf(theta + 0.5f * dTheta * gamma));
}
MIFX_D float env_mip_count(const EnvK& e) { return float(e.sphere ? e.smips : e.cube.mips); }

// ------------------------------------------------------------------------------------------------ I1
__global__ __launch_bounds__(256) void ibl_brdf_lut_kernel(Img out, unsigned numSamples)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= out.w || y >= out.h) return;
    const float NoV = (float(x) + 0.5f) / float(out.w), rough = (float(y) + 0.5f) / float(out.h);
    // IntegrateBRDF (PrecomputeBRDF.psh:8-39)
    const v3 V{sqrtf(1.0f - NoV * NoV), 0.0f, NoV};
    const v3 N{0.0f, 0.0f, 1.0f};
    float A = 0.0f, B = 0.0f;
    for (unsigned i = 0u; i < numSamples; ++i)
    {
        const v2 xi = hammersley2d(i, numSamples);
        const v3 H  = importance_sample_ggx(xi, rough, N);
        const v3 L  = 2.0f * dot(V, H) * H - V;
        const float NoL = saturate(L.z), NoH = saturate(H.z), VoH = saturate(dot(V, H));
        if (NoL > 0.0f)
        {
            const float alpha = rough * rough;
            const float gvis  = 4.0f * smith_ggx_visibility_correlated(NoL, NoV, alpha) * VoH * NoL / NoH;
            const float fc    = m_pow(1.0f - VoH, 5.0f);
            A += (1.0f - fc) * gvis;
            B += fc * gvis;
        }
    }
    st<v2>(out, x, y, v2{A / float(numSamples), B / float(numSamples)});
}

// ------------------------------------------------------------------------------------------------ I2 / I3
// SmithGGXSampleDirectionPDF (PBR_Common.fxh:297-324)
MIFX_D float smith_ggx_sample_direction_pdf(v3 V, v3 N, v3 L, float alpha)
{
    const v3    H = normalize(V + L);
    const float NdotH = dot(H, N), NdotV = dot(N, V), NdotL = dot(N, L);
    if (NdotH > 0.0f && NdotV > 0.0f && NdotL > 0.0f)
    {
        const float ndf  = normal_distribution_ggx(NdotH, alpha);
        const float g1   = smith_ggx_masking(NdotV, alpha);
        const float vndf = g1 * ndf / NdotV;
        return vndf / 4.0f;
    }
    return 0.0f;
}

__global__ __launch_bounds__(256) void ibl_prefilter_kernel(EnvK env, v4* out, int n, float roughness, unsigned numSamples)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= n || row >= 6 * n) return;
    const int face = row / n, y = row % n;
    const v3  R = normalize(cube_dir(face, (float(x) + 0.5f) / float(n), (float(y) + 0.5f) / float(n)));
    // PrefilterEnvMap (PrefilterEnvMap.psh:40-98), OPTIMIZE_SAMPLES = 1, ENV_MAP_TYPE_CUBE
    const v3 N = R, V = R;
    v3    color = mk3(0.0f);
    float total = 0.0f;
    const float mipCount = env_mip_count(env);
    for (unsigned i = 0u; i < numSamples; ++i)
    {
        const v2 xi = hammersley2d(i, numSamples);
        const v3 H  = importance_sample_ggx(xi, roughness, N);
        const v3 L  = 2.0f * dot(V, H) * H - V;
        const float NoL = clampf(dot(N, L), 0.0f, 1.0f), VoH = clampf(dot(V, H), 0.0f, 1.0f);
        if (NoL > 0.0f && VoH > 0.0f)
        {
            const float alpha  = roughness * roughness;
            const float pdf    = fmaxf(smith_ggx_sample_direction_pdf(V, N, L, alpha), 0.0001f);
            const float omegaS = 1.0f / (float(numSamples) * pdf);
            const float omegaP = env_pixel_solid_angle(env, L, 1.0f);
            const float mipLevel = (alpha == 0.0f) ? 0.0f : clampf(0.5f * m_log2(omegaS / fmaxf(omegaP, 1e-10f)) + 1.0f, 0.0f, mipCount - 1.0f);
            color += xyz(env_sample(env, L, mipLevel)) * NoL;
            total += NoL;
        }
    }
    out[size_t(row) * n + x] = mk4(color / total, 0.0f);
}

__global__ __launch_bounds__(256) void ibl_irradiance_kernel(EnvK env, v4* out, int n, unsigned numSamples)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= n || row >= 6 * n) return;
    const int face = row / n, y = row % n;
    const v3  N = normalize(cube_dir(face, (float(x) + 0.5f) / float(n), (float(y) + 0.5f) / float(n)));
    // IrradianceMap (ComputeIrradianceMap.psh:43-83): cosine-weighted hemisphere sampling with solid-angle based mip selection
    const v3 T = normalize(cross(N, fabsf(N.y) > 0.5f ? v3{1.0f, 0.0f, 0.0f} : v3{0.0f, 1.0f, 0.0f})); // BasisFromNormal (ShaderUtilities.fxh:104-110)
    const v3 B = cross(T, N);
    v3 irr = mk3(0.0f);
    const float mipCount = env_mip_count(env);
    for (unsigned i = 0u; i < numSamples; ++i)
    {
        const v2 xi = hammersley2d(i, numSamples);
        // SampleDirectionCosineHemisphere (PBR_Common.fxh:26-37)
        v3 L{m_cos(2.0f * MIFX_PI * xi.x) * sqrtf(1.0f - xi.y), m_sin(2.0f * MIFX_PI * xi.x) * sqrtf(1.0f - xi.y), sqrtf(xi.y)};
        const float pdf = fmaxf(L.z, 1e-6f) / MIFX_PI;
        L = normalize(L.x * T + L.y * B + L.z * N);
        const float omegaS = 1.0f / (float(numSamples) * pdf);
        const float omegaP = env_pixel_solid_angle(env, L, 0.5f);
        const float mipLevel = clampf(0.5f * m_log2(omegaS / fmaxf(omegaP, 1e-10f)) + 1.0f, 0.0f, mipCount - 1.0f);
        irr += xyz(env_sample(env, L, mipLevel));
    }
    out[size_t(row) * n + x] = mk4(irr / float(numSamples), 1.0f);
}

static mifx_status make_cubek(const mifx_cubemap* c, CubeK& k)
{
    MIFX_REQUIRE(c != nullptr && c->size > 0 && c->mip_count > 0 && c->mip_count <= 12 && (c->size >> (c->mip_count - 1)) >= 1, "environment map: bad cube map");
    k.size = int(c->size);
    k.mips = int(c->mip_count);
    for (uint32_t i = 0; i < 12; ++i) k.mip[i] = i < c->mip_count ? static_cast<const v4*>(c->mip_data[i]) : nullptr;
    for (uint32_t i = 0; i < c->mip_count; ++i) MIFX_REQUIRE(c->mip_data[i] != nullptr, "environment map: mip %u is null", i);
    return MIFX_OK;
}

// ------------------------------------------------------------------------------------------------ E1: environment-map background
// Shaders/Common/private/EnvMap.psh:46-77 (SampleEnvMap) behind EnvMap.vsh:9-23 and the pipeline state of Components/src/EnvMapRenderer.cpp:176-183:
// a full-screen triangle at the far plane with depth test LESS_EQUAL and no depth writes, i.e. every pixel whose depth is the far-plane depth
// gets the environment colour (cube sampled with a linear-clamp, mip-linear sampler at g_MipLevel, scaled, tone mapped, optionally gamma 2.2)
// and its motion vector (the direction is at infinity: only the rotation of the camera moves it); all other pixels are left as they are.
struct EnvMapK
{
    float farDepth, mipLevel, alpha;
    int   reversedDepth; // OPTION_FLAG_USE_REVERSE_DEPTH: COMPARISON_FUNC_GREATER_EQUAL (EnvMapRenderer.cpp:182)
    float scale[3];
    int   motionVectors;
};
template <int MODE, bool GAMMA>
__global__ __launch_bounds__(256) void envmap_kernel(EnvK env, Img depth, Img color, Img motion, CamK cam, CamK prev, EnvMapK k, ToneMapK tm)
{
    int x, y;
    if (!pixel_xy(color, x, y)) return;
    const float d = ld<float>(depth, x, y);
    if (!(k.reversedDepth ? k.farDepth >= d : k.farDepth <= d)) return; // the depth test of the far-plane triangle
    const float u = fdiv(float(x) + 0.5f, float(color.w)), v = fdiv(float(y) + 0.5f, float(color.h));
    const v4 clip{2.0f * u - 1.0f, 1.0f - 2.0f * v, k.farDepth, 1.0f}; // the interpolated CLIP_POS
    const v4 world = mul(clip, cam.viewProjInv);
    const v3 dir   = xyz(world) / world.w - v3{cam.pos[0], cam.pos[1], cam.pos[2]};
    v3 c = xyz(env_sample(env, normalize(dir), k.mipLevel)) * v3{k.scale[0], k.scale[1], k.scale[2]};
    if (MODE > 0) c = tone_map<MODE>(c, tm);
    if (GAMMA) c = pow3(c, 1.0f / 2.2f);
    st<v4>(color, x, y, mk4(c, k.alpha));
    if (motion.p != nullptr)
    {
        v2 mv{0.0f, 0.0f};
        if (k.motionVectors) // :61-66
        {
            const v3 prevWorld = v3{prev.pos[0], prev.pos[1], prev.pos[2]} + dir;
            const v4 prevClip  = mul(mk4(prevWorld, 1.0f), prev.viewProj);
            mv = v2{(clip.x - cam.jx) - (fdiv(prevClip.x, prevClip.w) - prev.jx), (clip.y - cam.jy) - (fdiv(prevClip.y, prevClip.w) - prev.jy)}; // GetMotionVector
        }
        st<v2>(motion, x, y, mv);
    }
}

static mifx_status make_envk(const mifx_cubemap* cube, const mifx_spheremap* sphere, EnvK& e)
{
    e = EnvK{};
    if (cube != nullptr) return make_cubek(cube, e.cube);
    MIFX_REQUIRE(sphere != nullptr && sphere->width > 0 && sphere->height > 0 && sphere->mip_count > 0 && sphere->mip_count <= 12, "environment map: bad sphere map");
    MIFX_REQUIRE((sphere->width >> (sphere->mip_count - 1)) >= 1 || (sphere->height >> (sphere->mip_count - 1)) >= 1, "environment map: %u mips for a %ux%u sphere map",
                 sphere->mip_count, sphere->width, sphere->height);
    for (uint32_t i = 0; i < sphere->mip_count; ++i)
    {
        MIFX_REQUIRE(sphere->mip_data[i] != nullptr, "environment map: sphere map mip %u is null", i);
        e.smip[i] = static_cast<const v4*>(sphere->mip_data[i]);
    }
    e.sw = int(sphere->width); e.sh = int(sphere->height); e.smips = int(sphere->mip_count); e.sphere = 1;
    return MIFX_OK;
}

mifx_status launch_envmap(hipStream_t s, const mifx_envmap_render_attribs& a, const mifx_tone_mapping_attribs& tm, const mifx_camera_attribs& cam, const mifx_camera_attribs& prev,
                          Img depth, Img color, Img motion)
{
    EnvK e;
    MIFX_CHECK(make_envk(a.env_map, a.sphere_map, e));
    const EnvMapK  k{cam.fFarPlaneDepth, a.mip_level, a.alpha, (a.options & MIFX_ENVMAP_OPTION_FLAG_USE_REVERSE_DEPTH) ? 1 : 0, {a.scale[0], a.scale[1], a.scale[2]}, (a.options & MIFX_ENVMAP_OPTION_FLAG_COMPUTE_MOTION_VECTORS) ? 1 : 0};
    const ToneMapK t = make_tonemapk(tm, a.average_log_lum);
    const CamK     c = make_camk(cam), p = make_camk(prev);
    const dim3 block(64, 4, 1);
    const dim3 grid = grid2d(color, block);
    const bool gamma = (a.options & MIFX_ENVMAP_OPTION_FLAG_CONVERT_OUTPUT_TO_SRGB) != 0;
#define MIFX_ENV_LAUNCH(M)                                                                                        \
    if (gamma) hipLaunchKernelGGL((envmap_kernel<M, true>), grid, block, 0, s, e, depth, color, motion, c, p, k, t); \
    else hipLaunchKernelGGL((envmap_kernel<M, false>), grid, block, 0, s, e, depth, color, motion, c, p, k, t)
    MIFX_TONEMAP_DISPATCH(tm.iToneMappingMode, MIFX_ENV_LAUNCH)
#undef MIFX_ENV_LAUNCH
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

mifx_status launch_ibl_brdf_lut(hipStream_t s, Img out, uint32_t num_samples)
{
    const dim3 block(64, 4, 1);
    hipLaunchKernelGGL(ibl_brdf_lut_kernel, grid2d(out.w, out.h, block), block, 0, s, out, num_samples);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_ibl_prefilter(hipStream_t s, const mifx_cubemap* env, const mifx_spheremap* sphere, void* out, uint32_t out_size, float roughness, uint32_t num_samples)
{
    EnvK e;
    MIFX_CHECK(make_envk(env, sphere, e));
    const dim3 block(32, 8, 1);
    hipLaunchKernelGGL(ibl_prefilter_kernel, grid2d(int(out_size), int(6 * out_size), block), block, 0, s, e, static_cast<v4*>(out), int(out_size), roughness, num_samples);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_ibl_irradiance(hipStream_t s, const mifx_cubemap* env, const mifx_spheremap* sphere, void* out, uint32_t out_size, uint32_t num_samples)
{
    EnvK e;
    MIFX_CHECK(make_envk(env, sphere, e));
    const dim3 block(32, 8, 1);
    hipLaunchKernelGGL(ibl_irradiance_kernel, grid2d(int(out_size), int(6 * out_size), block), block, 0, s, e, static_cast<v4*>(out), int(out_size), num_samples);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
