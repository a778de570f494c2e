"""Shared helpers of the parity tests (test infrastructure)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# parity contract (BASELINE.json north_star): 1e-3 relative per channel in fp32.  The absolute floor keeps the
# relative measure meaningful for values near zero (denormal-scale differences are not signal).
RTOL = 1e-3
ATOL = 1e-5


def blue_noise_tables():
    z = np.load(os.path.join(GOLDEN, "blue_noise_tables.npz"))
    return z["sobol_256d"], z["scrambling_tile"]


def rel_err(a, b, atol=ATOL):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), atol / RTOL)


def assert_close(got, want, rtol=RTOL, atol=ATOL, max_outlier_frac=0.0, what="", abs_slack=None, outlier_cap=None):
    """|got - want| <= rtol * max(|want|, atol/rtol) [+ abs_slack] for all but `max_outlier_frac` of the values.
    Outliers are only tolerated for passes with data-dependent discontinuities (documented at the call site); `outlier_cap` bounds what an
    outlier may be: no value (or, given as (magnitude, fraction), no more than that fraction of the values) may differ by more than
    magnitude * max(|want|, 1) -- a flipped decision moves a texel, it does not break the image.
    abs_slack: an array broadcastable to `got` with a derived, per-texel absolute allowance (documented at the call site)."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad_nan = np.isnan(got) != np.isnan(want)
    g = np.nan_to_num(got, nan=0.0, posinf=3e38, neginf=-3e38).astype(np.float64)
    w = np.nan_to_num(want, nan=0.0, posinf=3e38, neginf=-3e38).astype(np.float64)
    d = np.abs(g - w)
    if abs_slack is not None:
        d = np.maximum(d - np.asarray(abs_slack, np.float64), 0.0)
    e = d / np.maximum(np.abs(w), atol / RTOL)
    bad = (e > rtol) | bad_nan
    frac = float(bad.mean()) if bad.size else 0.0
    if os.environ.get("MIFX_PARITY_LOG"):  # tools/parity_table.py: the measured outlier fraction of every comparison of a run, beside its budget
        import json

        with open(os.environ["MIFX_PARITY_LOG"], "a") as fh:
            fh.write(json.dumps({"what": str(what), "frac": frac, "allowed": float(max_outlier_frac), "max_rel": float(e.max()) if e.size else 0.0, "n": int(bad.size)}) + "\n")
        if os.environ.get("MIFX_PARITY_MEASURE"):
            return float(e.max()) if e.size else 0.0, frac  # (developer mode: record, do not decide)
    assert frac <= max_outlier_frac, f"{what}: {bad.sum()} of {bad.size} values ({frac:.3e}) exceed rtol={rtol} (max rel err {e.max():.3e}, allowed frac {max_outlier_frac})"
    if outlier_cap is not None and bad.size:
        # outlier_cap = magnitude, or (magnitude, fraction of the values that may exceed it): the second tier of the budget
        cap, cap_frac = outlier_cap if isinstance(outlier_cap, tuple) else (outlier_cap, 0.0)
        big = d / np.maximum(np.abs(w), 1.0) > cap
        assert not bad_nan.any() and float(big.mean()) <= cap_frac, (f"{what}: {big.sum()} of {big.size} values ({big.mean():.3e}) differ by more than {cap} of max(|want|, 1) "
                                                                      f"(allowed {cap_frac}); worst {float((d / np.maximum(np.abs(w), 1.0)).max()):.3e}")
    return float(e.max()), frac


def centre_tap_slack(src, eps=None):
    """Allowance for passes that sample a same-size image at the texel centre (Bloom's `g_TextureInput.SampleLevel(CenterTexcoord)`,
    Bloom_ComputeUpsampledTexture.fx:44-52).  The checker evaluates that tap with the exact fp32 bilinear weights of the computed uv
    (GetBilinearSamplingInfoUC): the texel gets 1 - O(1e-5) and its neighbours O(1e-5), a rounding artefact of uv = ndc * 0.5 + 0.5 that a
    texture unit (8-bit weights) does not have; the HIP kernels load the texel itself.  The two differ by at most eps * the largest texel of
    the 3x3 neighbourhood -- invisible except beside a texel hundreds of times brighter.  eps = the rounding of uv (2^-23 per operation, a few
    operations) times the image size in texels.  Returns that bound per texel, (H, W, 1)."""
    a = np.abs(np.asarray(src, np.float64))
    if a.ndim == 3:
        a = a[..., :3].max(-1)
    if eps is None:
        eps = 2.0 * 2.0 ** -23 * max(a.shape)
    p = np.pad(a, 1, mode="edge")
    h, w = a.shape
    m = np.max([p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], axis=0)
    return (eps * m)[..., None]


def tone_mapping_attribs_bytes(mode, middle_gray=0.18, white_point=3.0, lum_sat=1.0, agx=(1.0, 1.0, 1.0, 0.0)):
    import struct

    return struct.pack("<iififfIIffff", mode, 1, middle_gray, 1, white_point, lum_sat, 0, 0, *agx)


def to_np(t):
    return t.detach().cpu().numpy()
