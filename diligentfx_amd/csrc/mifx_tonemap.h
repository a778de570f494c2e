// mifx_tonemap.h -- device implementation of the reference's ToneMap() for all TONE_MAPPING_MODE_* operators.
// Follows Shaders/PostProcess/ToneMapping/public/ToneMapping.fxh:8-226 and Shaders/Common/public/SRGBUtilities.fxh:4-38.
#pragma once
#include "mifx.h"
#include "mifx_device.h"

namespace mifx
{
struct ToneMapK // the fields ToneMap() reads, passed by value
{
    float middleGray, whitePoint, lumSaturation;
    float agxSaturation, agxSlope, agxPower, agxOffset;
    float aveLogLum;
    int   packedIn; // the input is Bloom's R11G11B10_FLOAT output plane (native-storage build; mifx_device.h: ld_hdr)
};

MIFX_D v3 srgb_to_linear(v3 c) // SRGBUtilities.fxh:4-10
{
    auto f = [](float s) {
        float less = s >= 0.04045f ? 1.0f : 0.0f; // step(0.04045, s)
        float lo   = s / 12.92f;
        float hi   = m_pow(saturate((s + 0.055f) / 1.055f), 2.4f);
        return lo + less * (hi - lo);
    };
    return v3{f(c.x), f(c.y), f(c.z)};
}
MIFX_D v3 linear_to_srgb(v3 c) // SRGBUtilities.fxh:27-33
{
    auto f = [](float s) {
        float gr = s >= 0.0031308f ? 1.0f : 0.0f;
        float lo = s * 12.92f;
        float hi = m_pow(s, 1.0f / 2.4f) * 1.055f - 0.055f;
        return lo + gr * (hi - lo);
    };
    return v3{f(c.x), f(c.y), f(c.z)};
}

MIFX_D v3 uncharted2(v3 x) // ToneMapping.fxh:8-19
{
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
MIFX_D v3 agx_contrast(v3 x) // ToneMapping.fxh:21-33
{
    v3 x2 = x * x;
    v3 x4 = x2 * x2;
    return 15.5f * x4 * x2 - 40.14f * x4 * x + 31.96f * x4 - 6.868f * x2 * x + 0.4298f * x2 + 0.1191f * x - 0.00232f;
}
MIFX_D v3 agx(v3 c) // ToneMapping.fxh:35-56
{
    const v3 r0{0.842479062253094f, 0.0784335999999992f, 0.0792237451477643f};
    const v3 r1{0.0423282422610123f, 0.878468636469772f, 0.0791661274605434f};
    const v3 r2{0.0423756549057051f, 0.0784336f, 0.879142973793104f};
    const float MinEv = -12.47393f, MaxEv = 4.026069f;
    c = v3{dot(r0, c), dot(r1, c), dot(r2, c)}; // mul(M, v): rows dotted with v
    c = v3{clampf(m_log2(c.x), MinEv, MaxEv), clampf(m_log2(c.y), MinEv, MaxEv), clampf(m_log2(c.z), MinEv, MaxEv)};
    c = (c - MinEv) / (MaxEv - MinEv);
    return agx_contrast(c);
}
MIFX_D v3 agx_eotf(v3 c) // ToneMapping.fxh:58-72
{
    const v3 r0{+1.19687900512017f, -0.0980208811401368f, -0.0990297440797205f};
    const v3 r1{-0.0528968517574562f, +1.15190312990417f, -0.0989611768448433f};
    const v3 r2{-0.0529716355144438f, -0.0980434501171241f, +1.15107367264116f};
    c = v3{dot(r0, c), dot(r1, c), dot(r2, c)};
    return srgb_to_linear(c);
}
MIFX_D v3 agx_look(v3 c, float sat, float offset, float slope, float power) // ToneMapping.fxh:74-85
{
    float lum = dot(c, v3{0.212671f, 0.715160f, 0.072169f});
    c = pow3(c * slope + offset, power);
    return lum + sat * (c - lum);
}

template <int MODE> MIFX_D v3 tone_map(v3 color, const ToneMapK& a) // ToneMapping.fxh:87-226
{
    const v3    lumW{0.212671f, 0.715160f, 0.072169f};
    const float lumScale = a.middleGray / a.aveLogLum;
    color                = max3(color, 0.0f);
    const float pixLum   = fmaxf(dot(lumW, color), 1e-10f);
    const float scaledLum = pixLum * lumScale;
    const v3    scaled   = color * lumScale;
    const float wp       = a.whitePoint;

    if constexpr (MODE == MIFX_TONE_MAPPING_MODE_EXP)
    {
        float t = 1.0f - m_exp(-scaledLum);
        return t * pow3(color / pixLum, a.lumSaturation);
    }
    else if constexpr (MODE == MIFX_TONE_MAPPING_MODE_REINHARD)
    {
        float t = scaledLum / (1.0f + scaledLum);
        return t * pow3(color / pixLum, a.lumSaturation);
    }
    else if constexpr (MODE == MIFX_TONE_MAPPING_MODE_REINHARD_MOD)
    {
        float t = scaledLum * (1.0f + scaledLum / (wp * wp)) / (1.0f + scaledLum);
        return t * pow3(color / pixLum, a.lumSaturation);
    }
    else if constexpr (MODE == MIFX_TONE_MAPPING_MODE_UNCHARTED2)
    {
        v3 curr  = uncharted2(2.0f * scaled);
        v3 white = mk3(1.0f) / uncharted2(mk3(wp));
        return curr * white;
    }
    else if constexpr (MODE == MIFX_TONE_MAPPING_MODE_FILMIC_ALU)
    {
        v3 t = max3(scaled - mk3(0.004f), 0.0f);
        t    = (t * (6.2f * t + mk3(0.5f))) / (t * (6.2f * t + mk3(1.7f)) + mk3(0.06f));
        return pow3(t, 2.2f);
    }
    else if constexpr (MODE == MIFX_TONE_MAPPING_MODE_LOGARITHMIC)
    {
        float t = m_log10(1.0f + scaledLum) / m_log10(1.0f + wp);
        return t * pow3(color / pixLum, a.lumSaturation);
    }
    else if constexpr (MODE == MIFX_TONE_MAPPING_MODE_ADAPTIVE_LOG)
    {
        const float Bias = 0.85f;
        float t = 1.0f / m_log10(1.0f + wp) * m_log(1.0f + scaledLum) / m_log(2.0f + 8.0f * m_pow(scaledLum / wp, m_log(Bias) / m_log(0.5f)));
        return t * pow3(color / pixLum, a.lumSaturation);
    }
    else if constexpr (MODE == MIFX_TONE_MAPPING_MODE_AGX)
    {
        return agx_eotf(agx(scaled));
    }
    else if constexpr (MODE == MIFX_TONE_MAPPING_MODE_AGX_CUSTOM)
    {
        v3 t = agx(scaled);
        t    = agx_look(t, a.agxSaturation, a.agxOffset, a.agxSlope, a.agxPower);
        return agx_eotf(t);
    }
    else if constexpr (MODE == MIFX_TONE_MAPPING_MODE_PBR_NEUTRAL)
    {
        color = color * (0.3f / a.aveLogLum);
        const float StartCompression = 0.8f - 0.04f, Desaturation = 0.15f;
        float x      = min_comp(color);
        float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
        color        = color - offset;
        float peak   = max_comp(color);
        if (peak >= StartCompression)
        {
            float d       = 1.0f - StartCompression;
            float newPeak = 1.0f - d * d / (peak + d - StartCompression);
            color         = color * (newPeak / peak);
            float g       = 1.0f - 1.0f / (Desaturation * (peak - newPeak) + 1.0f);
            color         = lerp3(color, mk3(newPeak), g);
        }
        return color;
    }
    else if constexpr (MODE == MIFX_TONE_MAPPING_MODE_COMMERCE)
    {
        color = color * (0.3f / a.aveLogLum);
        const float StartCompression = 0.8f, Desaturation = 0.5f;
        float d    = 1.0f - StartCompression;
        float peak = max_comp(color);
        if (peak >= StartCompression)
        {
            float newPeak = 1.0f - d * d / (peak + d - StartCompression);
            float invPeak = 1.0f / peak;
            float extra   = dot(color * (1.0f - StartCompression * invPeak), mk3(1.0f));
            color         = color * (newPeak * invPeak);
            float g       = 1.0f - 3.0f / (Desaturation * extra + 3.0f);
            color         = lerp3(color, mk3(1.0f), g);
        }
        return color;
    }
    else
    {
        return color; // TONE_MAPPING_MODE_NONE: after the max(color, 0) above, as in the reference
    }
}

inline ToneMapK make_tonemapk(const mifx_tone_mapping_attribs& a, float ave_log_lum)
{
    return ToneMapK{a.fMiddleGray, a.fWhitePoint, a.fLuminanceSaturation, a.AgXSaturation, a.AgXSlope, a.AgXPower, a.AgXOffset, ave_log_lum};
}

// run-time dispatch over the compile-time mode (mode is validated by the caller)
#define MIFX_TONEMAP_DISPATCH(mode, FN)                 \
    switch (mode)                                       \
    {                                                   \
        case 0: FN(0); break;  case 1: FN(1); break;    \
        case 2: FN(2); break;  case 3: FN(3); break;    \
        case 4: FN(4); break;  case 5: FN(5); break;    \
        case 6: FN(6); break;  case 7: FN(7); break;    \
        case 8: FN(8); break;  case 9: FN(9); break;    \
        case 10: FN(10); break; case 11: FN(11); break; \
        default: break;                                 \
    }

} // namespace mifx
