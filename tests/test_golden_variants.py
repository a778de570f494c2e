"""Golden fixture of the remaining permutations (tests/golden/variants_golden.npz: outputs of the reference compiled for the CPU, see make_golden_variants.py):
PCF-shadowed shade, previous-frame SSR, equirectangular IBL precompute + background, auto exposure.  The hand-written oracle reproduces them from the stored
inputs -- the pin of the oracle for these rows where oracle/_ref is not available."""
import importlib.util
import os

import numpy as np

from util import GOLDEN, assert_close


def test_oracle_reproduces_golden_variants(oracle):
    spec = importlib.util.spec_from_file_location("_make_golden_variants", os.path.join(GOLDEN, "make_golden_variants.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    data = np.load(os.path.join(GOLDEN, "variants_golden.npz"))
    got = gen.run(oracle, "oracle_", data, np.load(os.path.join(GOLDEN, "next_golden.npz")))
    # thresholded decisions (a PCF tap on "reference < texel", a ray crossing a tile) may flip on isolated texels between two fp32 builds of the same arithmetic
    outliers = {"out_shadowed_radiance_pcf3": 1e-3, "out_shadowed_radiance_pcf7": 1e-3, "out_ssr_previous_frame0": 4e-3, "out_ssr_previous_frame1": 4e-3,
                "out_sphere_prefiltered": 1e-3, "out_sphere_irradiance": 1e-3}
    assert sorted(got) == sorted(k for k in data.files if k.startswith("out_")) and len(got) == 10
    for name, value in got.items():
        assert_close(value, data[name], rtol=2e-4, atol=1e-6, max_outlier_frac=outliers.get(name, 0.0), what=f"golden {name}")
    # the fixture exercises what it claims to: shadows darken, the previous-frame hits differ from the current-frame ones, the sun is in the background
    nxt = np.load(os.path.join(GOLDEN, "next_golden.npz"))
    assert (data["out_shadowed_radiance_pcf3"] != data["out_shadowed_radiance_pcf7"]).any()
    assert (data["out_ssr_previous_frame1"] != nxt["fwd1_out_ssr"]).any()
    assert data["out_sphere_background"][..., :3].max() > 20.0 and len(set(np.round(data["out_autoexposure_averages"], 5))) >= 3
