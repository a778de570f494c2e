"""Frame runner used by bench.py: builds the synthetic inputs of one GPU's share of the frame, steps the chain, times the passes.

world == 1: the whole frame on one GPU.
world  > 1: every rank renders its own view of --width x --height pixels (weak scaling); see sharding_note()."""
import os
import time

import numpy as np
import torch

from . import api, binding as B, synth

ALGO_BPP = {"pbr_shade": 84.0, "prep": 28.0, "ssr": 327.7, "ssao": 148.0, "composite": 116.0, "taa": 64.0, "bloom": 74.7, "tonemap": 32.0}


class TiledChain:
    def __init__(self, device_index, sobol, tile, rank, world, width, height):
        self.rank, self.world, self.w, self.h = rank, world, width, height
        self.tables = (sobol, tile)
        self.chain = api.Chain(device_index, sobol, tile)
        self.dev = self.chain.device
        self.scene = synth.Scene()
        self.frames = []
        self.out = None

    def sharding_note(self):
        if self.world == 1:
            return "single GPU, whole frame"
        return f"{self.world} GPUs, one {self.w}x{self.h} view per GPU (independent frames, no data-path collective)"

    # ------------------------------------------------------------------ inputs
    def build_inputs(self, n_frames=2, first_frame=16):
        """G-buffers of `n_frames` consecutive camera positions (steady state: frames >= 16); the bench alternates between them
        while the frame index keeps increasing, so every temporal pass takes its history path."""
        w, h, dev = self.w, self.h, self.dev
        # each rank looks at the scene from its own orbit phase (rank-dependent first frame) so that the ranks do not render identical pixels
        base = first_frame + 40 * self.rank
        self.frames = [synth.make_frame(self.scene, base + i, w, h, dev) for i in range(n_frames)]
        env = synth.make_sky_cube(256, dev)
        self.ibl = api.precompute_ibl(self.chain.postfx, env)  # reference defaults: LUT 512^2/512, irradiance 64^2/8192, prefiltered 256^2 x 9 mips/256
        self.shade = synth.make_lights()
        self.shade.PrefilteredCubeLastMip = float(len(self.ibl.pre) - 1)
        self.out = torch.empty(h, w, 4, device=dev)
        self.bound = [None] * n_frames
        torch.cuda.synchronize(dev)

    def step(self, i):
        """One frame of the chain. Frame indices are consecutive (history is kept); the G-buffer alternates between the resident frames."""
        k = i % len(self.frames)
        b = self.chain.bind_frame(1000 + i, self.frames[k], self.ibl, self.shade, self.out)
        self.chain.execute(b)

    # ------------------------------------------------------------------ per-pass timing (HIP events on the launch stream)
    def time_passes(self, reps=10):
        """Times every pass group separately through the same C ABI the chain uses (steady state), `reps` launches each."""
        import ctypes

        dev, w, h = self.dev, self.w, self.h
        f = self.frames[0]
        ctx = api.PostFXContext(dev.index or 0, *self.tables)
        ssao, ssr, taa, bloom = api.ScreenSpaceAmbientOcclusion(ctx), api.ScreenSpaceReflection(ctx), api.TemporalAntiAliasing(ctx), api.Bloom(ctx)
        g = {k: f[k] for k in ("base_color", "normal", "material", "depth")}
        rad = torch.empty(h, w, 4, device=dev)
        spec = torch.empty(h, w, 4, device=dev)
        comp = torch.empty(h, w, 4, device=dev)
        final = torch.empty(h, w, 4, device=dev)
        tm = B.ToneMappingAttribs.default(4)
        res = {}
        frame_no = [5000]

        def prepare():
            frame_no[0] += 1
            ctx.prepare_resources(frame_no[0], w, h)
            ssao.prepare_resources()
            ssr.prepare_resources()
            taa.prepare_resources(api.TemporalAntiAliasing.FEATURE_FLAG_BICUBIC_FILTER)
            bloom.prepare_resources()

        def run_prep():
            ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])

        stages = {
            "pbr_shade": lambda: api.pbr_shade(ctx, g, f["camera"], self.shade, self.ibl, out_radiance=rad, out_specular_ibl=spec),
            "prep": run_prep,
            "ssr": lambda: ssr.execute(rad, f["depth"], f["normal"], f["material"], f["motion"], B.SSRAttribs.default()),
            "ssao": lambda: ssao.execute(f["depth"], f["normal"], B.SSAOAttribs.default()),
            "composite": lambda: api.composite(ctx, rad, spec, ssr.get_ssr_radiance(), ssao.get_ambient_occlusion(), f["normal"], f["base_color"], f["material"],
                                               self.ibl.lut, f["camera"], out=comp),
            "taa": lambda: taa.execute(comp, B.TAAAttribs.default()),
            "bloom": lambda: bloom.execute(taa.get_accumulated_frame(), B.BloomAttribs.default()),
            "tonemap": lambda: ctx.tone_map(bloom.get_bloom_texture(), tm, 0.3, flags=1, out=final),
        }
        # warm the whole sequence a few frames so that histories exist
        for _ in range(6):
            prepare()
            for fn in stages.values():
                fn()
        torch.cuda.synchronize(dev)
        acc = {k: 0.0 for k in stages}
        for _ in range(reps):
            prepare()
            for name, fn in stages.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                acc[name] += e0.elapsed_time(e1)
        for name in stages:
            res[name] = {"ms": acc[name] / reps, "algo_bytes": ALGO_BPP[name] * w * h}
        for o in (bloom, taa, ssr, ssao):
            o.close()
        ctx.close()
        return res
