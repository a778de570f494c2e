"""Golden fixtures (tests/golden/chain_golden.npz, outputs of the reference compiled for the CPU, see make_golden_chain.py):
  * CPU: the hand-written oracle reproduces them;
  * GPU: the HIP chain reproduces them through the C ABI."""
import numpy as np
import pytest

import chain_util
import cpu_chain
from util import assert_close


def test_oracle_reproduces_golden_chain(oracle):
    ibl, frames, sa = chain_util.load_golden()
    chain = cpu_chain.CpuChain(oracle, "oracle_")
    for i, fr in enumerate(frames):
        keep = {}
        final = chain_util.run_frame_inputs(chain, fr["in"], fr["camera"], fr["prev_camera"], i, ibl, sa, keep)
        keep["final"] = final
        for name, want in fr["out"].items():
            assert_close(keep[name], want, rtol=2e-4, atol=1e-6, max_outlier_frac=2e-3, what=f"golden frame {i} {name}")


@pytest.mark.gpu
def test_hip_chain_reproduces_golden(mifx_lib):
    import torch

    from diligentfx_amd import api, binding as B
    from util import blue_noise_tables, to_np

    ibl_np, frames, sa = chain_util.load_golden()
    sobol, tile = blue_noise_tables()
    chain = api.Chain(0, sobol, tile)
    dev = chain.device
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(dev), [torch.from_numpy(m).to(dev) for m in ibl_np["irradiance"]],
                           [torch.from_numpy(m).to(dev) for m in ibl_np["prefiltered"]])
    for i, fr in enumerate(frames):
        g = {k: torch.from_numpy(v).to(dev) for k, v in fr["in"].items()}
        g["camera"], g["prev_camera"] = B.camera_from_bytes(fr["camera"]), B.camera_from_bytes(fr["prev_camera"])
        h, w = fr["in"]["depth"].shape
        out = torch.zeros(h, w, 4, device=dev)
        chain.execute(chain.bind_frame(i, g, ibl, sa, out))
        # stochastic rays / thresholded history decisions may flip on isolated texels; the image must otherwise agree to 1e-3
        assert_close(to_np(out), fr["out"]["final"], max_outlier_frac=2.5e-4, what=f"golden final frame {i}")  # (measured 1.22e-4 on an MI355X, round 4)
        assert np.abs(to_np(out) - fr["out"]["final"]).mean() < 1e-3
    chain.close()
