"""GPU parity on the configuration that bench.py times: many CONSECUTIVE frames of a real orbit, so that the saturated-history paths run --
SSAO history length at SSAO_MAX_HISTORY_LENGTH = 16 (ScreenSpaceAmbientOcclusionStructures.fxh:57), A7 on its early-out
(SSAO_ComputeResampledHistory.fx:56-64), R6 / T1 on their steady blends (TemporalAntiAliasing.cpp:123-143) -- compared with the checker at
frames 1, 2, 8, 17 and beyond (SURVEY.md section 4 iv).

Two kinds of comparison:
  * end to end: the HIP chain and the CPU chain run independently from a reset; decisions that flip in one of them (SSR rays, history
    rejection) travel through the temporal filters, so a bounded fraction of texel-channels may differ -- by a bounded amount;
  * one frame from a saturated history: the CPU chain's history after N frames is imported into the HIP effects
    (mifx_*_import_history), both sides run frame N + 1: what differs is one frame's arithmetic, not N frames of divergence."""
import os

import numpy as np
import pytest
import torch

import cpu_chain
from util import assert_close, blue_noise_tables, to_np

pytestmark = pytest.mark.gpu

W, H = 384, 216
FRAMES = 24
CHECK = (0, 1, 7, 16, 19, 23)  # 0-based: frames 1, 2, 8, 17 (SURVEY 4 iv), 20, 24

# Outlier budgets = 2 - 2.5 x the fractions measured on an MI355X at frame 17 (final 1.9e-3, SSAO 5.0e-3, SSR 5.7e-3, TAA 5.0e-3; they grow
# from 4e-4 / 6e-4 / 2.7e-3 / 1.7e-3 at frame 1 as flipped decisions travel through the temporal filters; profiles/r02_steady_state_parity.txt),
# with a cap on what an outlier may be: 5e-2 of max(|want|, 1) (measured worst 2.4e-2) -- a flipped ray or history rejection moves a texel,
# it does not break it.
BUDGET_FINAL, BUDGET_SSAO, BUDGET_SSR, BUDGET_TAA = 5e-3, 1.2e-2, 1.2e-2, 1.2e-2  # measured (round 4, no-contraction build): 2.5e-3 / 6.7e-3 / 5.9e-3 / 6.1e-3
CAP = (5e-2, 2e-4)  # the second tier: at most 2e-4 of the values (a few dozen texel-channels here) may be off by more than 5e-2 -- an SSR ray that lands on the
                    # sun's reflection on one side only changes its pixel completely (measured: 0.22 on one texel of frame 21)
CAP_AO = 0.5  # an AO texel whose history is accepted on one side and rejected on the other jumps between its accumulated and its one-frame value (measured 0.18)
ONE_FRAME = {"ssao": 5e-4, "ssr": 1.5e-3, "taa": 9e-4, "final": 3.6e-4}  # measured 2.5e-4 / 7.4e-4 / 4.3e-4 / 1.8e-4
if os.environ.get("MIFX_PARITY_MEASURE"):  # developer mode: report the fractions without deciding (how the budgets above were obtained)
    BUDGET_FINAL = BUDGET_SSAO = BUDGET_SSR = BUDGET_TAA = 1.0
    CAP = CAP_AO = None
    ONE_FRAME = dict.fromkeys(ONE_FRAME, 1.0)


def checker():
    import pyref

    r = pyref.ref_lib()
    if r is not None:
        return r, "ref_"
    o = pyref.oracle_lib()
    if not o.has("oracle_pbr_shade"):
        pytest.skip("no checker with the full chain available")
    return o, "oracle_"


def setup_chain():
    import chain_util
    from diligentfx_amd import api, synth

    lib, pfx = checker()
    sobol, tile = blue_noise_tables()
    chain = api.Chain(0, sobol, tile)
    ibl_np = chain_util.make_ibl(lib, pfx)
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(chain.device), [torch.from_numpy(m).to(chain.device) for m in ibl_np["irradiance"]],
                           [torch.from_numpy(m).to(chain.device) for m in ibl_np["prefiltered"]])
    cpu = cpu_chain.CpuChain(lib, pfx)
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    return chain, cpu, ibl, ibl_np, sa, synth.Scene()


def test_chain_24_consecutive_frames(mifx_lib):
    import chain_util
    from diligentfx_amd import synth

    chain, cpu, ibl, ibl_np, sa, scene = setup_chain()
    out = torch.zeros(H, W, 4, device=chain.device)
    report = []
    for frame in range(FRAMES):
        f = synth.make_frame(scene, frame, W, H, chain.device)
        chain.execute(chain.bind_frame(frame, f, ibl, sa, out))
        keep = {}
        want = chain_util.run_frame(cpu, scene, frame, W, H, ibl_np, keep)
        got = to_np(out)
        assert np.isfinite(got).all()
        if frame not in CHECK:
            continue
        fr = {}
        _, fr["final"] = assert_close(got, want, max_outlier_frac=BUDGET_FINAL, outlier_cap=CAP, what=f"final image, frame {frame + 1}")
        _, fr["ssao"] = assert_close(to_np(chain.effect_output("ssao")), keep["ssao_out"], max_outlier_frac=BUDGET_SSAO, outlier_cap=CAP_AO, what=f"SSAO, frame {frame + 1}")
        # SSR radiance is HDR (the sun's reflection reaches 1e2): the cap is relative to the value
        _, fr["ssr"] = assert_close(to_np(chain.effect_output("ssr")), keep["ssr_out"], max_outlier_frac=BUDGET_SSR, what=f"SSR, frame {frame + 1}")
        _, fr["taa"] = assert_close(to_np(chain.effect_output("taa")), keep["taa_out"], max_outlier_frac=BUDGET_TAA, what=f"TAA, frame {frame + 1}")
        assert np.abs(got[..., :3] - want[..., :3]).mean() < 1e-3  # and the images are the same picture
        hist_len = to_np(chain.effect("ssao").get_intermediate("history_len"))
        geom = to_np(f["depth"]) < 1.0 - 1e-6
        fr["worst"] = float((np.abs(got - want) / np.maximum(np.abs(want), 1.0)).max())
        fr["len16"] = float((hist_len[geom] >= 16.0).mean())
        fr["a7_early_out"] = float(((hist_len[geom] - 1.0) / 4.0 >= 1.0).mean())
        report.append((frame + 1, fr))
        print(f"frame {frame + 1:2d}: " + " ".join(f"{k} {v:.2e}" for k, v in fr.items()), flush=True)
        if frame >= 16:  # the timed configuration: histories at SSAO_MAX_HISTORY_LENGTH, A7 on its early-out for most surface pixels (at this
            # small size a pixel is a large solid angle: the 0.5 deg / frame orbit resets more histories than at 3840x2160)
            assert hist_len.max() == 16.0 and fr["len16"] > 0.2 and fr["a7_early_out"] > 0.7, fr
            # (the history length is reprojected bilinearly: fractional values, compared with a tolerance)
            assert (np.abs(hist_len - keep["ssao_hist_len"]) > 0.05).mean() < 2e-2
    chain.close()


def test_one_frame_from_saturated_history(mifx_lib):
    """CPU chain for 20 frames; its SSAO / SSR / TAA history goes into the HIP effects (import_history); frame 21 on both sides."""
    import chain_util
    from diligentfx_amd import synth

    chain, cpu, ibl, ibl_np, sa, scene = setup_chain()
    n = 20
    for frame in range(n):
        chain_util.run_frame(cpu, scene, frame, W, H, ibl_np)
    out = torch.zeros(H, W, 4, device=chain.device)
    # one HIP frame allocates the effect planes (prepare); then its history is replaced by the checker's
    f = synth.make_frame(scene, n - 1, W, H, chain.device)
    chain.execute(chain.bind_frame(n - 1, f, ibl, sa, out))
    dev = chain.device
    slot = (n - 1) & 1
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    chain.effect("ssao").import_history(t(cpu.ssao_hist["ao"][slot]), t(cpu.ssao_hist["len"][slot]), n - 1)
    # SSR keeps state outside the reflection mask as well (R5's targets and the history slot of the frame before last: ssr.hip): the complete state is the slot of frame n - 2
    # (an import with that index fills it), the slot of frame n - 1, and the three targets of R5, which are reachable as intermediates
    chain.effect("ssr").import_history(t(cpu.ssr_hist["rad"][slot ^ 1]), t(cpu.ssr_hist["var"][slot ^ 1]), n - 2)
    chain.effect("ssr").import_history(t(cpu.ssr_hist["rad"][slot]), t(cpu.ssr_hist["var"][slot]), n - 1)
    for name, plane in zip(("res_radiance", "res_variance", "res_depth"), cpu.ssr_hist["res"]):
        chain.effect("ssr").get_intermediate(name).copy_(t(plane))
    chain.effect("taa").import_history(t(cpu.taa_hist[slot]), n - 1)
    # round trip: export returns what was imported, bit for bit, with the frame index
    ao, ln, idx = chain.effect("ssao").export_history()
    assert idx == n - 1 and np.array_equal(to_np(ao), cpu.ssao_hist["ao"][slot]) and np.array_equal(to_np(ln), cpu.ssao_hist["len"][slot])
    rad, var, idx = chain.effect("ssr").export_history()
    assert idx == n - 1 and np.array_equal(to_np(rad), cpu.ssr_hist["rad"][slot]) and np.array_equal(to_np(var), cpu.ssr_hist["var"][slot])
    col, idx = chain.effect("taa").export_history()
    assert idx == n - 1 and np.array_equal(to_np(col), cpu.taa_hist[slot])

    f = synth.make_frame(scene, n, W, H, chain.device)
    chain.execute(chain.bind_frame(n, f, ibl, sa, out))
    keep = {}
    want = chain_util.run_frame(cpu, scene, n, W, H, ibl_np, keep)
    got = to_np(out)
    # one frame of arithmetic on identical history: the budgets are those of the per-pass tests (rays that flip, thresholds), far below
    # the end-to-end ones
    res = {}
    _, res["ssao"] = assert_close(to_np(chain.effect_output("ssao")), keep["ssao_out"], max_outlier_frac=ONE_FRAME["ssao"], outlier_cap=CAP_AO, what="SSAO from imported history")
    _, res["ssr"] = assert_close(to_np(chain.effect_output("ssr")), keep["ssr_out"], max_outlier_frac=ONE_FRAME["ssr"], what="SSR from imported history")
    _, res["taa"] = assert_close(to_np(chain.effect_output("taa")), keep["taa_out"], max_outlier_frac=ONE_FRAME["taa"], what="TAA from imported history")
    _, res["final"] = assert_close(got, want, max_outlier_frac=ONE_FRAME["final"], outlier_cap=CAP, what="final image from imported history")
    print("one frame from the checker's saturated history: outlier fractions " + " ".join(f"{k} {v:.2e}" for k, v in res.items()))
    hist_len = to_np(chain.effect("ssao").get_intermediate("history_len"))
    assert hist_len.max() == 16.0
    # export / import refuse what they cannot do
    from diligentfx_amd import binding as B

    chain.reset_history()
    with pytest.raises(B.MifxError, match="INVALID_OP"):
        chain.effect("ssao").export_history()
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        chain.effect("taa").import_history(torch.zeros(H, W + 1, 4, device=dev), 3)
    chain.close()
