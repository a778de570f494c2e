"""diligentfx_amd -- MI355X-native implementation of the DiligentFX per-pixel hot path (PBR shade + PostProcess chain).

The product is diligentfx_amd/libmifx.so (hand-written HIP kernels for gfx950 behind the C ABI of include/mifx.h).
This package only binds it; it has no CPU / PyTorch fallback and raises when the library is missing."""
from . import binding  # noqa: F401

__all__ = ["binding", "api", "load"]


def load():
    return binding.load()
