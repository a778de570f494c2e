// TEST INFRASTRUCTURE ONLY (oracle/_ref).  C2: Shaders/Common/private/ComputeReprojectedDepth.fx (ComputeReprojectedDepthPS :18),
// host wiring PostProcess/Common/src/PostFXContext.cpp:611-633.
#include "ref_common.h"
namespace hlsl { namespace c2 {
#include "ShaderDefinitions.fxh"
#include "ComputeReprojectedDepth.fx"
}}
using namespace hlsl;

// in[0]: depth; cam0/cam1; out[0]: reprojected depth
extern "C" int ref_reprojected_depth(const ref_args* a)
{
    ref_bind(c2::g_TextureDepth.s, a, 0);
    std::memcpy(&c2::g_CurrCamera, a->cam0, sizeof(c2::CameraAttribs));
    std::memcpy(&c2::g_PrevCamera, a->cam1, sizeof(c2::CameraAttribs));
    const ref_img& o = a->out[0];
    ref_fullscreen<c2::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](c2::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, c2::ComputeReprojectedDepthPS(vs)); });
    return 0;
}
extern "C" int ref_sizeof_camera_attribs() { return int(sizeof(c2::CameraAttribs)); }
