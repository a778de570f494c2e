#!/usr/bin/env python3
"""Generates tests/golden/layers_golden.npz: outputs of the REFERENCE (oracle/_ref/libmifx_ref.so: PBR_Shading.fxh / PBR_Common.fxh / Iridescence.fxh compiled per permutation,
oracle/ref/ref_pl_*.cpp) for the shade with material layers -- every single layer, all five, and all five with two shadow-mapped lights -- on one small frame.  The fixture
carries its own inputs (G-buffer, layer planes, the two sheen tables, IBL maps, shadow maps, camera, attribs), so that the tests that read it need neither /root/reference nor
oracle/_ref.  Run in the build container:   python tests/golden/make_golden_layers.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

SIZE = (56, 36)
CASES = [("clearcoat", False), ("sheen", False), ("anisotropy", True), ("iridescence", False), ("transmission", False), ("all", True), ("all_shadows3", True)]  # (permutation, optional planes bound)


def main():
    import torch

    import chain_util
    import pyref
    from layers_util import checker_result, make_case

    ref = pyref.ref_lib()
    assert ref is not None and ref.has("ref_pbr_shade_layers_all_shadows3"), "oracle/_ref with the layered permutations is needed (python -c 'import __graft_entry__ as g; g.build()')"
    ibl = chain_util.make_ibl(ref, "ref_", env_size=16, lut_size=16, irr_size=4, pref_size=8, lut_samples=32, irr_samples=64, pref_samples=16)
    d = {"ibl_lut": ibl["lut"], "ibl_irradiance": ibl["irradiance"][0]}
    for m, p in enumerate(ibl["prefiltered"]):
        d[f"ibl_prefiltered{m}"] = p
    # one frame and one set of layer planes for all cases; lights with and without shadow maps
    f, gn, sa_shadowed, planes, albedo, charlie = make_case("golden", SIZE, ibl, torch.device("cpu"), shadowed=True)
    _, _, sa_plain, _, _, _ = make_case("golden", SIZE, ibl, torch.device("cpu"), shadowed=False)
    slices, infos = chain_util.make_shadow_inputs(size=32)
    for k, v in gn.items():
        d["g_" + k] = v
    for k, v in planes.items():
        d["layer_" + k] = v
    d["lut_albedo_scaling"], d["lut_charlie"] = albedo, charlie
    d["camera"] = np.frombuffer(bytes(f["camera"]), np.uint8)
    d["shade_attribs"] = np.frombuffer(bytes(sa_plain), np.uint8)
    d["shade_attribs_shadowed"] = np.frombuffer(bytes(sa_shadowed), np.uint8)
    d["shadow_slices"], d["shadow_infos"] = np.stack(slices), infos
    for perm, optional in CASES:
        shadowed = "shadows" in perm
        r, s = checker_result(ref, perm, optional, f, gn, sa_shadowed if shadowed else sa_plain, planes, albedo, charlie, ibl, shadows=(slices, infos) if shadowed else None)
        assert np.isfinite(r).all() and np.isfinite(s).all()
        d[f"out_{perm}_radiance"], d[f"out_{perm}_specular_ibl"] = r, s
    out = os.path.join(HERE, "layers_golden.npz")
    np.savez_compressed(out, **d)
    print(out, os.path.getsize(out), "bytes,", len(d), "arrays")


if __name__ == "__main__":
    main()
