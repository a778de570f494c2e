#!/usr/bin/env python3
"""Folds the per-kernel FETCH_SIZE / WRITE_SIZE summaries (tools/pmc_stats.py) into HBM bytes per frame per chain stage.

    python tools/pmc_traffic.py pmc_FETCH_SIZE.txt pmc_WRITE_SIZE.txt frames > profiles/rNN_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): the counters are KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled.
Both are calibrated in the same run on a streaming kernel with known bytes (the tone map, or the final Bloom up-sample that carries it since round 2)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAGE = {"pbr_shade": "pbr_shade", "cube_apron": "pbr_shade", "composite": "composite", "taa": "taa", "tonemap": "tonemap", "blue_noise": "prep", "postfx_prep": "prep",
         "ssr_": "ssr", "ssao_": "ssao", "bloom_": "bloom"}


def parse(path):
    """tools/pmc_stats.py table: kernel ... | dispatches | dur_us | counter per dispatch (KiB) -> KiB summed over all dispatches"""
    out = {}
    for line in open(path).read().splitlines()[1:]:
        f = line.split()
        name = " ".join(f[:-3]).replace("mifx::", "")
        out[name] = float(f[-1]) * int(f[-3])
    return out


def main():
    fetch, write, frames = parse(sys.argv[1]), parse(sys.argv[2]), int(sys.argv[3])
    stages, kernels = {}, {}
    for name in fetch:
        st = next((s for k, s in STAGE.items() if name.startswith(k)), None)
        if st is None:
            continue
        rd, wr = 2.0 * fetch[name] * 1024 / frames, write.get(name, 0.0) * 1024 / frames
        kernels[name] = {"read_bytes": round(rd), "write_bytes": round(wr)}
        stages[st] = stages.get(st, 0.0) + rd + wr
    # calibration kernel: the streaming pass with known bytes -- the tone map (16 B/px in, 16 out) or, since round 2 fused it into Bloom's final up-sample,
    # that kernel (colour 16 + the quarter-size up-sampled level 4 in; Bloom output 16 + LDR frame 16 out)
    px = 3840 * 2160
    h4 = len(sys.argv) > 5 and sys.argv[5] == "h4"  # the RGBA16_FLOAT storage build: 4-channel texels are 8 bytes
    c4 = 8 if h4 else 16
    tm = next((k for k in kernels if k.startswith("tonemap")), None)
    exp = (c4 * px, c4 * px)
    if tm is None:
        # (round 4: with MIFX_CHAIN_FUSE_BLOOM_OUTPUT_ON_DEMAND -- the default -- the kernel writes the LDR frame only; BLOOM_OUT_WRITTEN=1 in the environment for a run
        #  with --fusion-mask 15, where it also writes the Bloom output plane)
        import os

        writes = 2 if os.environ.get("BLOOM_OUT_WRITTEN") == "1" else 1
        tm, exp = next(k for k in kernels if k.startswith("bloom_final_tonemap")), ((c4 + c4 // 4) * px, (c4 if h4 and writes == 1 else 16 if writes == 1 else 2 * c4) * px)
    # the device code these counters were taken from (round 6): bench.py reports the traffic only for a library with the same kernels
    from diligentfx_amd import binding as B

    lib = os.environ.get("MIFX_LIB_PATH") or (os.path.join(ROOT, "diligentfx_amd", "libmifx_h4.so") if h4 else B.LIB_PATH)
    print(json.dumps({"resolution": [3840, 2160], "frames": frames, "unit": "bytes per frame", "fetch_correction": 2.0, "build": sys.argv[4] if len(sys.argv) > 4 else "",
                      "device_code_sha16": B.device_code_sha16(lib), "library": os.path.relpath(lib, ROOT), "storage": "RGBA16_FLOAT 4-channel planes" if h4 else "fp32 planes",
                      "calibration": {"kernel": tm, "expected_read": exp[0], "expected_write": exp[1], **kernels[tm]},
                      "stage_traffic": {k: round(v) for k, v in stages.items()}, "chain_traffic": round(sum(stages.values())),
                      "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
