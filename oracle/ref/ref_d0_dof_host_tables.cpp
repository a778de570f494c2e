// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D0: the host-side tables of the depth-of-field effect, GenerateKernelPoints and GenerateGaussKernel
// (PostProcess/DepthOfField/src/DepthOfField.cpp:49-94), taken from the file where it lies (ref_prep.py EXTRACTS; the rest of that file needs
// DiligentCore).  Int32 and PI_F come from DiligentCore (un-vendored submodule: Primitives/interface/BasicTypes.h, Common/interface/BasicMath.hpp:
// PI_F = 3.14159265358979323846f).
#include "ref_common.h"
#include <vector>
namespace hlsl { namespace d0 {
using Int32 = int;
static constexpr float PI_F = 3.14159265358979323846f;
#include "dof_host_extract.inc"
}}
using namespace hlsl;

// out[0]: n x 1 kernel points (c=2), zero padded; ival[0] = ring count, ival[1] = ring density
extern "C" int ref_dof_kernel_points(const ref_args* a)
{
    const ref_img& o = a->out[0];
    std::vector<float2> k = d0::GenerateKernelPoints(a->ival[0], a->ival[1]);
    if (o.c != 2 || o.h != 1 || size_t(o.w) < k.size()) return -1;
    std::memset(o.data, 0, sizeof(float) * 2 * size_t(o.w));
    for (size_t i = 0; i < k.size(); ++i) { o.data[2 * i] = k[i].x; o.data[2 * i + 1] = k[i].y; }
    return 0;
}
// out[0]: (2 * radius + 1) x 1 weights; ival[0] = radius, fval[0] = sigma
extern "C" int ref_dof_gauss_kernel(const ref_args* a)
{
    const ref_img& o = a->out[0];
    std::vector<float> k = d0::GenerateGaussKernel(a->ival[0], a->fval[0]);
    if (o.c != 1 || o.h != 1 || size_t(o.w) != k.size()) return -1;
    std::memcpy(o.data, k.data(), sizeof(float) * k.size());
    return 0;
}
