// mifx_interop.hpp -- the two functions the APPLICATION provides so that DiligentFX effects backed by mifx can see its textures.
// A texture must live in memory HIP can address: a Vulkan image exported with VK_KHR_external_memory and imported with
// hipImportExternalMemory (linear tiling), a D3D12 / Vulkan buffer shared the same way and bound as a texture view on the graphics side,
// or planes that are HIP allocations to begin with (a HIP-native G-buffer).
#pragma once
#include <mifx.h>
#include "diligent_api_stand_in.hpp" // in the reference tree: "TextureView.h"

namespace Diligent
{
// device pointer + pitch of the view's texture; `Format` is the MIFX_FORMAT_* the caller expects (checked against the texture's own format)
mifx_image2d GetMifxImage(ITextureView* pView, uint32_t Format);
// the reverse: a texture view over an effect-owned plane (valid until the next PrepareResources that changes size or flags)
ITextureView* WrapMifxImage(const mifx_image2d& Image);
// a cube-map SRV (all mips) as the stacked-face float4 mip chain mifx_cubemap describes -- the prefiltered environment and the irradiance map of PBR_Renderer
// (GetPrefilteredEnvMapSRV / GetIrradianceCubeSRV, PBR_Renderer.hpp); false when the view cannot be shared
bool GetMifxCubemap(ITextureView* pCubeView, mifx_cubemap& Cube);
// the HIP stream the graphics queue is synchronised with (external semaphores), nullptr = the default stream
void* GetMifxStream(IDeviceContext* pContext);
} // namespace Diligent
