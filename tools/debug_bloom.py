import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/oracle", "/root/repo/tests"]
import numpy as np, torch
import cpu_chain, pyref
from util import blue_noise_tables, to_np, rel_err
from diligentfx_amd import api, binding as B, synth
import test_gpu_attribute_sweeps as T
lib, pfx = T.checker()
a = B.BloomAttribs.default()
a.Intensity, a.Threshold, a.SoftTreshold, a.Radius, a.AlphaInterpolation = 0.05, 2.0, 0.0, 1.0, 0.4
for frame, f, bloom, chain, pf in T.drive(lambda api, ctx: api.Bloom(ctx)):
    color = (T.scene_color(f) * 3.0).contiguous()
    bloom.execute(color, a)
    keep = {}
    want = chain.bloom(to_np(color), a, keep)
    got = to_np(bloom.get_bloom_texture())
    def rep(name, g, w):
        e = rel_err(g, w)
        bad = e > 1e-3
        print(f"frame {frame} {name} {g.shape}: max rel {e.max():.3e} bad {bad.sum()}/{bad.size}", flush=True)
        if bad.any():
            idx = np.argwhere(bad)[:6]
            for i in idx:
                print("    ", tuple(i), g[tuple(i)], w[tuple(i)])
    for i, d in enumerate(keep["bloom_down"]):
        rep(f"down{i}", to_np(bloom.get_intermediate(f"down{i}")), d)
    for i, u in enumerate(keep["bloom_up"]):
        rep(f"up{i}", to_np(bloom.get_intermediate(f"up{i}")), u)
    rep("out", got, want)
    # isolate: feed checker with GPU's own intermediates for each up level
    n = len(keep["bloom_down"])
    for i in range(n - 1, 0, -1):
        src = to_np(bloom.get_intermediate(f"up{i}")) if i != n - 1 else to_np(bloom.get_intermediate(f"down{i}"))
        o = np.zeros_like(keep["bloom_up"][i - 1])
        lib.call(pfx + "bloom_upsample", [to_np(bloom.get_intermediate(f"down{i-1}")), src], [o], attribs=bytes(a), ival=[0])
        rep(f"iso-up{i-1}", to_np(bloom.get_intermediate(f"up{i-1}")), o)
    for i in range(1, n):
        o = np.zeros_like(keep["bloom_down"][i])
        lib.call(pfx + "bloom_downsample", [to_np(bloom.get_intermediate(f"down{i-1}"))], [o])
        rep(f"iso-down{i}", to_np(bloom.get_intermediate(f"down{i}")), o)
    o = np.zeros_like(keep["bloom_down"][0])
    lib.call(pfx + "bloom_prefilter", [to_np(color)], [o], attribs=bytes(a))
    rep("iso-down0", to_np(bloom.get_intermediate("down0")), o)
