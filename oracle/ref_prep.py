#!/usr/bin/env python3
"""ref_prep.py -- TEST INFRASTRUCTURE (build step of oracle/_ref, never part of the product).

Reads the reference's HLSL shader sources where they lie under /root/reference (read-only), applies a
handful of purely syntactic rewrites so that g++ accepts them through oracle/ref/hlsl_shim.h, and writes
the result into a *temporary* build directory (never into this repository).  It also generates the
swizzle member lists used by the shim.  No reference source is copied into the repo or its history.

Rewrites (HLSL syntax with no C++ spelling; semantics untouched):
  * comments stripped (newlines kept, so line numbers still match the reference file:line)
  * `Texture2D<T>` -> `Texture2D_<T>`, bare `Texture2D` -> `Texture2D_<float4>`
  * parameter qualifiers: `in T x` -> `T x`, `out T x` / `inout T x` -> `T& x` (arrays: `T (&x)[N]`)
  * semantics `: SV_Target0`, `: SV_Position`, ... removed; `[unroll]`-style attributes removed
"""
import itertools
import os
import re
import sys

REFERENCE_DIRS = [
    "Shaders/Common/public",
    "Shaders/Common/private",
    "Shaders/PostProcess/ToneMapping/public",
    "Shaders/PostProcess/ScreenSpaceAmbientOcclusion/public",
    "Shaders/PostProcess/ScreenSpaceAmbientOcclusion/private",
    "Shaders/PostProcess/ScreenSpaceReflection/public",
    "Shaders/PostProcess/ScreenSpaceReflection/private",
    "Shaders/PostProcess/TemporalAntiAliasing/public",
    "Shaders/PostProcess/TemporalAntiAliasing/private",
    "Shaders/PostProcess/Bloom/public",
    "Shaders/PostProcess/Bloom/private",
    "Shaders/PostProcess/DepthOfField/public",
    "Shaders/PostProcess/DepthOfField/private",
    "Shaders/PBR/public",
    "Shaders/PBR/private",
]
EXTS = (".fx", ".fxh", ".psh")

_comment_re = re.compile(r"//[^\n]*|/\*.*?\*/", re.S)


def _strip_comments(src: str) -> str:
    def repl(m):
        return "".join(ch if ch == "\n" else " " for ch in m.group(0))

    return _comment_re.sub(repl, src)


_TYPE = r"[A-Za-z_]\w*(?:\s*<[^<>]*>)?"


def transform(src: str) -> str:
    s = _strip_comments(src)
    s = re.sub(r"\bTexture2DArray\s*<", "Texture2DArray_<", s)
    s = re.sub(r"\bTexture2D\s*<", "Texture2D_<", s)
    s = re.sub(r"\bTexture2D\b(?!_)", "Texture2D_<float4>", s)
    # attributes
    s = re.sub(r"\[\s*(unroll|loop|earlydepthstencil|branch|flatten)\s*(\([^)]*\))?\s*\]", "", s)
    # semantics
    s = re.sub(r":\s*(SV_\w+|NORMALIZED_XY|INSTANCE_ID|WORLD_POS)\b", "", s)

    # out / inout parameters -> references
    def ref_param(m):
        ty, name, arr = m.group(2), m.group(3), m.group(4)
        if arr:
            return f"{ty} (&{name}){arr}"
        return f"{ty}& {name}"

    s = re.sub(r"\b(inout|out)\s+(" + _TYPE + r")\s+(\w+)\s*(\[[^\]]*\])?", ref_param, s)
    # in parameters: drop the qualifier (only when followed by a type-looking token inside a parameter list)
    s = re.sub(r"([(,]\s*)in\s+(?=[A-Za-z_])", r"\1", s)
    s = re.sub(r"^(\s*)in\s+(?=[A-Za-z_][\w<>]*\s+\w+\s*[,)\n])", r"\1", s, flags=re.M)  # parameter on its own line (after an #endif)
    return s


def gen_swizzles(outdir: str) -> None:
    for n in (2, 3, 4):
        lines = []
        for names in ("xyzw"[:n], "rgba"[:n]):
            for k in (2, 3, 4):
                for combo in itertools.product(range(n), repeat=k):
                    nm = "".join(names[i] for i in combo)
                    idx = ",".join(str(i) for i in combo)
                    lines.append(f"swz<vec<T,{k}>,T,HL_N,{idx}> {nm};")
        with open(os.path.join(outdir, f"hlsl_swizzles_{n}.inc"), "w") as f:
            f.write("\n".join(lines) + "\n")


def _extract_definition(src: str, start: str) -> str:
    """The text of the definition that begins with `start` up to its matching closing brace (or the end of the line for a #define)."""
    i = src.index(start)
    if start.startswith("#define"):
        return src[i:src.index("\n", i) + 1]
    if start.rstrip().endswith("="):  # a statement (`const float X =` ... `;`) out of a function that cannot be compiled whole
        return src[i:src.index(";", i) + 1] + "\n"
    j = src.index("{", i)
    depth = 0
    while True:
        depth += {"{": 1, "}": -1}.get(src[j], 0)
        j += 1
        if depth == 0:
            return src[i:j] + "\n"


# Auto exposure (ref/ref_n3_autoexposure.cpp) needs three definitions out of the light-scattering post-process; the files they live in include
# that whole effect, so only these definitions are taken from the files where they lie.
EXTRACTS = {
    "epls_autoexposure_extract.inc": [
        ("Shaders/PostProcess/EpipolarLightScattering/private/AtmosphereShadersCommon.fxh", "#define RGB_TO_LUMINANCE"),
        ("Shaders/PostProcess/EpipolarLightScattering/private/AtmosphereShadersCommon.fxh", "float2 GetWeightedLogLum("),
        ("Shaders/PostProcess/EpipolarLightScattering/private/UpdateAverageLuminance.fx", "void UpdateAverageLuminancePS("),
    ],
    # Depth of field (ref/ref_d0_dof_host_tables.cpp): the two host-side table generators, out of a file that otherwise needs DiligentCore
    "dof_host_extract.inc": [
        ("PostProcess/DepthOfField/src/DepthOfField.cpp", "static std::vector<float2> GenerateKernelPoints("),
        ("PostProcess/DepthOfField/src/DepthOfField.cpp", "static std::vector<float> GenerateGaussKernel("),
    ],
    # Host helpers of TemporalAntiAliasing and ToneMapping (ref/ref_t0_host_helpers.cpp): self-contained arithmetic out of files that otherwise need DiligentCore.
    # GetJitterOffset is a member function that looks its buffer up in a map first; its two arithmetic statements are taken on their own.
    "taa_host_extract.inc": [
        ("PostProcess/TemporalAntiAliasing/src/TemporalAntiAliasing.cpp", "static float HaltonSequence("),
        ("PostProcess/TemporalAntiAliasing/interface/TemporalAntiAliasing.hpp", "static inline float4x4 GetJitteredProjMatrix("),
    ],
    "taa_jitter_statements_extract.inc": [
        ("PostProcess/TemporalAntiAliasing/src/TemporalAntiAliasing.cpp", "constexpr Uint32 SampleCount ="),
        ("PostProcess/TemporalAntiAliasing/src/TemporalAntiAliasing.cpp", "const float      JitterX     ="),
        ("PostProcess/TemporalAntiAliasing/src/TemporalAntiAliasing.cpp", "const float      JitterY     ="),
    ],
    "tonemap_host_extract.inc": [
        ("Components/src/ToneMapping.cpp", "float3 ReverseExpToneMap("),
    ],
}


def main(ref_root: str, outdir: str) -> int:
    os.makedirs(outdir, exist_ok=True)
    gen_swizzles(outdir)
    for name, parts in EXTRACTS.items():
        text = ""
        for rel, start in parts:
            with open(os.path.join(ref_root, rel), "r", encoding="utf-8", errors="replace") as f:
                text += _extract_definition(_strip_comments(f.read()), start)
        with open(os.path.join(outdir, name), "w") as f:
            f.write(transform(text))
    with open(os.path.join(outdir, "PSMainGenerated.generated"), "w") as f:
        f.write("// EnvMap.psh includes the application's pixel-shader main here (EnvMapRenderer.cpp:96-103); the wrapper calls SampleEnvMap itself\n")
    seen = {}
    for d in REFERENCE_DIRS:
        full = os.path.join(ref_root, d)
        if not os.path.isdir(full):
            print(f"ref_prep: missing {full}", file=sys.stderr)
            return 1
        for fn in sorted(os.listdir(full)):
            if not fn.endswith(EXTS):
                continue
            if fn in seen:
                print(f"ref_prep: duplicate basename {fn} ({seen[fn]} vs {d})", file=sys.stderr)
                return 1
            seen[fn] = d
            with open(os.path.join(full, fn), "r", encoding="utf-8", errors="replace") as f:
                src = f.read()
            with open(os.path.join(outdir, fn), "w") as f:
                f.write(f'#line 1 "{os.path.join(full, fn)}"\n')
                f.write(transform(src))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
