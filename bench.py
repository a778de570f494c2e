#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native DiligentFX hot path on synthetic G-buffers.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one frame of the full chain (BASELINE.json configs[3]): PBR GGX+IBL shade -> PostFX prep -> SSR -> SSAO -> composite ->
TAA -> Bloom -> ToneMap at 3840x2160 per GPU, steady state (temporal history warmed up).  Inputs (the G-buffers of a pre-rendered camera
orbit of consecutive frames -- tiling.TiledChain.build_inputs -- and the IBL maps) are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

Roofline accounting: algorithmic bytes = every distinct input texel read once + every output texel written
once per reference pass in fp32 storage; 874.3 B/px for the whole chain.  `roofline` reports the dominant kernel of the frame (longest
duration, found by bench.py itself in an untimed sweep over the bracketed kernels), every launch of it inside the timed region measured
with HIP events on the launch stream; `roofline.lowest` is the kernel furthest below the HBM roofline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured achievable copy rate

# algorithmic bytes per pixel per pass group
ALGO_BPP = {"pbr_shade": 84.0, "prep": 28.0, "ssr": 327.7, "ssao": 148.0, "composite": 116.0, "taa": 64.0, "dof": 0.0, "bloom": 74.7, "tonemap": 32.0}
CHAIN_BPP = sum(ALGO_BPP.values())
# The dominant kernel of the chain (largest total time in profiles/r01_kernel_stats_*.txt): the SSR ray march R4.  Algorithmic bytes per pixel
# (SURVEY Appendix C, R4): reads normal 16 + roughness 4 + depth pyramid 5.33 + mask 1 + radiance at the hit 16, writes 2 x float4 = 32.
# Depth of field (--dof), fp32 planes, per full-resolution pixel: D1 8 + D2 20 + D3/D4 5.3 + D5 0.1 + D6 28.1 + D7 16 + D8 16 + D9 16 + D10 40
DOF_BPP = 149.5
DOF_LENS = (12.0, 1.2, 135.0)  # focus distance (m), f-stop, focal length (mm)
# Algorithmic bytes per pixel of the kernels whose launch sites carry a HIP-event bracket (per reference pass).  bench.py
# times each of them over a few untimed frames, quotes `roofline` on the one with the longest duration and `roofline.lowest` on the one
# furthest below the HBM roofline -- no kernel name is hard-wired.
KERNEL_BPP = {"pbr_shade_kernel": 84.0, "pbr_shade_ssr_mask_kernel": 84.0 + 25.0, "bloom_upsample_tonemap_kernel": 36.0 + 32.0,  # (fused kernels: the sum of the reference passes they perform)
              "postfx_prep_kernel": 28.0, "ssr_mask_roughness_kernel": 25.0, "ssr_intersection_kernel": 74.33, "ssr_spatial_kernel": 81.0,
              "ssr_temporal_kernel": 81.0, "ssr_bilateral_kernel": 61.0, "ssao_compute_ao_kernel": 25.33, "ssao_temporal_kernel": 36.0, "ssao_resample_kernel": 34.67,
              "ssao_spatial_kernel": 36.0, "ssao_resolve_list_kernels": 34.67 + 36.0, "composite_kernel": 116.0, "composite_ssr_cleanup_kernel": 116.0 + 61.0, "taa_kernel": 64.0, "bloom_prefilter_kernel": 20.0, "bloom_upsample_kernel": 36.0,
              "tonemap_kernel": 32.0}


# --storage h4 (the native-storage build, libmifx_h4.so; NOT the headline configuration): the same accounting (every distinct texel once) with the
# reference's own target formats -- 4-channel colour planes RGBA16_FLOAT (8 B), ambient occlusion and SSR roughness R8_UNORM (1 B), SSAO history length / SSR variance /
# SSR resolved depth R16_FLOAT (2 B), closest motion RG16_FLOAT (4 B), Bloom levels R11G11B10_FLOAT (4 B); depth, the depth pyramids, the reflection mask and the
# motion input keep 4 / 8 bytes.  469.3 B/px for the whole chain (fp32 storage: 874.3).
ALGO_BPP_H4 = {"pbr_shade": 44.0, "prep": 24.0, "ssr": 181.67, "ssao": 80.0, "composite": 57.0, "taa": 36.0, "dof": 0.0, "bloom": 30.67, "tonemap": 16.0}
KERNEL_BPP_H4 = {"pbr_shade_kernel": 44.0, "pbr_shade_ssr_mask_kernel": 44.0 + 14.0, "bloom_upsample_tonemap_kernel": 17.0 + 16.0, "postfx_prep_kernel": 24.0,
                 "ssr_mask_roughness_kernel": 14.0, "ssr_intersection_kernel": 39.33, "ssr_spatial_kernel": 42.0, "ssr_temporal_kernel": 49.0, "ssr_bilateral_kernel": 32.0,
                 "ssao_compute_ao_kernel": 14.33, "ssao_temporal_kernel": 19.0, "ssao_resample_kernel": 17.67, "ssao_spatial_kernel": 17.0, "ssao_resolve_list_kernels": 17.67 + 17.0, "composite_kernel": 57.0, "composite_ssr_cleanup_kernel": 57.0 + 32.0,
                 "taa_kernel": 36.0, "bloom_prefilter_kernel": 9.0, "bloom_upsample_kernel": 17.0, "tonemap_kernel": 16.0}


# ---------------------------------------------------------------- CPU baseline (the checker on the host cores; never the product path)
def usable_cores():
    """(threads to use, description): the host cores this process may actually run on -- physical cores, clipped by the scheduler affinity
    mask and by a cgroup CPU quota (a container often reports every core of the machine in os.cpu_count() but is throttled to a few)."""
    logical = os.cpu_count() or 1
    try:
        import psutil

        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = logical
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            f = open(path).read().split()
            if path.endswith("cpu.max"):
                if f[0] != "max":
                    quota = float(f[0]) / float(f[1])
            elif int(f[0]) > 0:
                quota = int(f[0]) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except Exception:
            continue
        break
    n = max(1, min(physical, affinity, int(quota + 0.5) if quota else physical))
    return n, f"{logical} logical / {physical} physical cores, affinity {affinity}, cgroup quota {('%.1f' % quota) if quota else 'none'}"


def pin_host_threads():
    """OpenMP settings of the CPU baseline, set before any OpenMP runtime loads (libgomp reads them once): one thread per usable physical
    core, bound close.  Only when the caller has not chosen otherwise."""
    n, _ = usable_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(n))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")


def cpu_baseline(budget_s=20.0, size=(3840, 2160), device=None):
    """Times the checker library running the same chain at the bench's own frame size (BASELINE configs[3]: 3840x2160) on the host cores:
    one warm-up frame (reset history), then consecutive frames of the same orbit until ~budget_s of CPU work, at least one.  The inputs are
    rendered before the timed region (on the GPU when one is given).  kind = "reference" when oracle/_ref travelled (the reference's shader
    source compiled for the CPU), "port" for the hand-written oracle."""
    import torch

    from diligentfx_amd import synth

    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import chain_util
    import cpu_chain
    import pyref

    lib, pfx, kind = pyref.ref_lib(), "ref_", "reference"
    if lib is None:
        lib, pfx, kind = pyref.oracle_lib(), "oracle_", "port"
        if not lib.has("oracle_ssr_intersection"):
            raise RuntimeError("no CPU checker with the full chain available")
    w, h = size
    ibl = chain_util.make_ibl(lib, pfx, env_size=64, lut_size=64, irr_size=16, pref_size=32, lut_samples=64, irr_samples=128, pref_samples=32)
    cpu = cpu_chain.CpuChain(lib, pfx)
    scene = synth.Scene()
    frames = list(range(16, 16 + 8))
    t_total, n = 0.0, 0
    pre = {}
    orig = synth.make_frame
    gen_dev = device if device is not None else torch.device("cpu")

    def cached(scene_, idx, w_, h_, dev_, rows=None, **kw):
        key = (idx, w_, h_)
        if key not in pre:  # rendered where it is fast, handed to the checker as host arrays
            f = orig(scene_, idx, w_, h_, gen_dev, rows, **kw)
            pre[key] = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in f.items()}
        return pre[key]

    synth.make_frame = cached
    try:
        cached(scene, frames[0], w, h, None)
        chain_util.run_frame(cpu, scene, frames[0], w, h, ibl)  # warm-up (page-in, history reset)
        for fr in frames[1:]:
            cached(scene, fr, w, h, None)  # outside the timed region
            t0 = time.perf_counter()
            chain_util.run_frame(cpu, scene, fr, w, h, ibl)
            t_total += time.perf_counter() - t0
            n += 1
            pre.pop((fr - 1, w, h), None)
            if t_total > budget_s:
                break
    finally:
        synth.make_frame = orig
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or usable_cores()[0]
    return {"value": round(w * h * n / t_total / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": kind,
            "sample": f"{n} consecutive frame(s) of the full chain at {w}x{h} after one warm-up frame, {t_total:.1f} s of CPU work "
                      f"({'oracle/_ref: reference shader source compiled for the CPU' if kind == 'reference' else 'oracle/mifx_oracle.cpp'}, OpenMP, "
                      f"OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')} OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}; host: {usable_cores()[1]})"}


def pmc_traffic(w, h, kernel):
    """HBM bytes per launch of `kernel` (and per frame of the whole chain) from the committed PMC passes -- counters cannot be read inside a
    timed run; (None, None) when no measurement exists for this resolution."""
    import glob

    suffix = "_h4" if os.environ.get("MIFX_STORAGE") == "h4" else ""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{suffix}.json")))  # the latest round's measurement of this storage build
    if not paths:
        return None, None, None
    t = json.load(open(paths[-1]))
    if t["resolution"] != [w, h]:
        return None, None, None
    k = next((v for name, v in t["kernels"].items() if name.startswith(kernel)), None)
    return (k["read_bytes"] + k["write_bytes"]) if k else None, t["chain_traffic"], os.path.relpath(paths[-1], ROOT)


def measured_copy_peak(dev, torch):
    """Achievable HBM rate of this device (SURVEY 8d asks for it beside the 8 TB/s spec): a 1 GiB device-to-device copy, read + write bytes."""
    n = 1 << 28
    a, b = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    a.fill_(1.0)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * 4.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=40)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160, help="rows per GPU (weak scaling: every GPU renders a full --width x --height view)")
    p.add_argument("--shard-rows", action="store_true", help="(the default for N > 1) the ranks share ONE frame by row bands with RCCL exchanges")
    p.add_argument("--verify-shard", action="store_true", help="sharded mode over torch.distributed: every rank also runs the unsharded chain on EVERY frame (inside the timed "
                   "region) and compares its band bit for bit; the default check runs a few extra frames after the timed region instead")
    p.add_argument("--comm", default="rccl", choices=("rccl", "torch"), help="sharded mode: exchanges inside libmifx over RCCL (default) or driven from Python over torch.distributed")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo with --single-gpu exercises the multi-rank code on one GPU)")
    p.add_argument("--single-gpu", action="store_true", help="testing: every rank uses cuda:0")
    p.add_argument("--dof", action="store_true", help="also run the depth-of-field effect (SURVEY 8f N1) between TAA and Bloom, temporal smoothing on, with a lens "
                   "that blurs both fields of the synthetic scene; not the BASELINE headline configuration")
    p.add_argument("--ssao-half", action="store_true", help="SSAO with FEATURE_FLAG_HALF_RESOLUTION (checkerboard depth, AO at half size, bilateral upsampling); not the headline configuration")
    p.add_argument("--ssr-half", action="store_true", help="SSR with FEATURE_FLAG_HALF_RESOLUTION (half-size mask and ray pass); not the headline configuration")
    p.add_argument("--orbit-frames", type=int, default=24, help="consecutive camera positions of the pre-rendered orbit resident in HBM (68 B/px each); the run walks them forwards and back")
    p.add_argument("--replicas", action="store_true", help="N > 1: every rank renders its own --width x --height view (weak scaling, no collective) instead of the default "
                   "for N > 1, ONE frame of 2*width x 2*height row-band sharded over the ranks (BASELINE configs[4])")
    p.add_argument("--storage", default="fp32", choices=("fp32", "h4"), help="h4: the native-storage build of the library (the reference's own target formats: RGBA16_FLOAT colour planes, "
                   "R8_UNORM / R16_FLOAT / RG16_FLOAT / R11G11B10_FLOAT for the narrow ones); a second configuration, not the fp32 headline")
    p.add_argument("--fusion-mask", type=lambda v: int(v, 0), default=None, help="A/B: mifx_chain_set_fusion_mask (MIFX_CHAIN_FUSE_*; default: every fusion on; 3 = round 2's chain)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-pass-breakdown", action="store_true")
    p.add_argument("--no-kernel-sweep", action="store_true", help="profiling runs (rocprofv3 counts frames): skip the untimed per-kernel sweep; the line then carries no `roofline`")
    return p.parse_args()


def kernel_sweep(runner, frames=3):
    """Untimed: every bracketed kernel in turn is timed with HIP events over `frames` consecutive frames -> {kernel: average ms per launch}."""
    times = {}
    for name in KERNEL_BPP:
        runner.arm_kernel_timing(name, frames)
        for _ in range(frames):
            runner.step()
        kt = runner.kernel_times_ms(frames)
        if kt:
            times[name] = sum(kt) / len(kt)
    runner.arm_kernel_timing(None, 0)
    return times


def main():
    args = parse_args()
    global KERNEL_BPP, ALGO_BPP, CHAIN_BPP
    if args.storage == "h4":
        os.environ["MIFX_STORAGE"] = "h4"  # read when diligentfx_amd.binding is imported
        KERNEL_BPP, ALGO_BPP = KERNEL_BPP_H4, ALGO_BPP_H4
        CHAIN_BPP = sum(ALGO_BPP.values())
        args.no_cpu_baseline = True  # the CPU baseline belongs to the fp32 headline
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        pin_host_threads()  # before any OpenMP runtime is loaded
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus or world == 1, (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from diligentfx_amd import api, binding as B, synth
    from diligentfx_amd import tiling

    tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
    W, H = args.width, args.height
    # N > 1: ONE frame of twice the width and height of the per-GPU configuration (7680x4320 = BASELINE configs[4]) sharded by row bands over
    # the ranks -- north_star's tile-parallel frame; --replicas keeps the N independent views of round 1
    shard = world > 1 and not args.replicas
    if shard and not args.shard_rows and (args.width, args.height) == (3840, 2160):
        W, H = 2 * args.width, 2 * args.height
    if shard and args.orbit_frames == 24:
        args.orbit_frames = 8  # 2.3 GB per resident 8K frame and rank

    # ---------------------------------------------------------------- inputs (resident in HBM before timing)
    runner = tiling.TiledChain(local_rank, tables["sobol_256d"], tables["scrambling_tile"], rank, world, W, H, shard_rows=shard, verify=args.verify_shard, comm_backend=args.comm)
    shared_frame = runner.shard_rows
    runner.build_inputs(n_frames=args.orbit_frames)
    chain_bpp = CHAIN_BPP
    tiling.ALGO_BPP.update(ALGO_BPP)
    if args.fusion_mask is not None:
        runner.chain.set_fusion_mask(args.fusion_mask)
    if args.ssao_half or args.ssr_half:
        assert not shared_frame, "--ssao-half / --ssr-half: not covered by the row-band phases"
        runner.chain.set_effect_feature_flags(ssao_feature_flags=2 if args.ssao_half else 0, ssr_feature_flags=2 if args.ssr_half else 0)
    if args.dof:
        assert not shared_frame, "--dof: the row-band phases do not cover the depth-of-field passes"
        for f in runner.frames:
            f["camera"].fFocusDistance, f["camera"].fFStop, f["camera"].fFocalLength = DOF_LENS
        runner.chain.set_depth_of_field(B.DOFAttribs.default(), api.DepthOfField.FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING)
        tiling.ALGO_BPP["dof"] = DOF_BPP
        chain_bpp += DOF_BPP

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        runner.step()
    # which kernel is the frame's longest, and which is furthest below its roofline: measured here, not assumed (every rank steps the same
    # number of frames: the sharded mode exchanges data inside step())
    ktimes = kernel_sweep(runner) if not args.no_kernel_sweep else {}
    dominant = max(ktimes, key=ktimes.get) if ktimes else None
    if rank == 0 and dominant:
        runner.arm_kernel_timing(dominant, args.steps)  # HIP events around every launch of the dominant kernel inside the timed region
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        runner.step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    from diligentfx_amd.dist import max_over_ranks

    elapsed = max_over_ranks(elapsed, dev)  # the slowest rank defines the step time (covered by tests/test_dist_gloo.py)

    total_px = float(W) * H * (1 if shared_frame else world) * args.steps
    value = total_px / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    rows_gpu = H // world if shared_frame else H  # output rows per GPU (ghost rows of the sharded mode are overhead, not work)
    chain_gbs = chain_bpp * W * rows_gpu / (dev_ms / args.steps * 1e-3) / 1e9

    result = {
        "metric": "Mpixels/s full PBR+postFX chain @4K; %HBM roofline; 1/2/4/8-GPU scaling",
        "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if shared_frame else "weak", "vs_baseline": None,
        "dtype": "f32" if args.storage == "fp32" else "f32 arithmetic, storage in the reference's target formats (RGBA16F / R8 / R16F / RG16F / R11G11B10F)", "data": "synthetic",
        "config": {"workload": (f"full chain PBR+SSR+SSAO+composite+TAA+Bloom+ToneMap, one {W}x{H} frame row-band sharded over {world} GPUs (BASELINE configs[4] layout)"
                                if shared_frame else f"full chain PBR+SSR+SSAO+composite+TAA+{'DOF+' if args.dof else ''}Bloom+ToneMap {W}x{H} per GPU (BASELINE configs[3]{' + depth of field' if args.dof else ''})"), "width": W, "height_per_gpu": rows_gpu,
                   "sharding": runner.sharding_note(), "storage": "fp32 planes" if args.storage == "fp32" else "libmifx_h4.so: RGBA16_FLOAT colour planes, R8_UNORM AO / roughness, R16_FLOAT variance / history length, RG16_FLOAT closest motion, R11G11B10_FLOAT Bloom levels; fp32 depth",
                   "taa": "bicubic", "ssao": "GTAO half-res + bilateral upsampling" if args.ssao_half else "GTAO full-res", "ssr": "half-res rays" if args.ssr_half else "full-res rays", "tonemap": "Uncharted2+sRGB",
                   "chain_algorithmic_bytes_per_px": round(chain_bpp, 1), "chain_hbm_frac": round(chain_gbs / HBM_PEAK_GBS, 4)},
    }

    # ---------------------------------------------------------------- roofline of the dominant kernel (HIP events over the timed region)
    if rank == 0 and dominant:
        kt = runner.kernel_times_ms(args.steps)
        runner.arm_kernel_timing(None, 0)
        k_ms = sum(kt) / max(len(kt), 1)
        px = W * rows_gpu

        def roof(name, ms):
            algo = KERNEL_BPP[name] * px
            ach = algo / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            traffic, chain_traffic, src = pmc_traffic(W, H, name) if not shared_frame else (None, None, None)
            return {"kernel": name, "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": round(algo),
                    "kernel_ms": round(ms, 5)}, chain_traffic, src

        dom, chain_traffic, src = roof(dominant, k_ms)
        fracs = {n: KERNEL_BPP[n] * px / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS for n, ms in ktimes.items() if ms > 0}
        lowest_name = min(fracs, key=fracs.get)
        lowest, _, _ = roof(lowest_name, ktimes[lowest_name])
        lowest["measured"] = "untimed sweep, 3 launches"
        copy_gbs = measured_copy_peak(dev, torch)
        result["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": dom["frac"], "traffic": dom["traffic"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                              "kernel_ms": dom["kernel_ms"], "launches_timed": len(kt),
                              "traffic_source": (src + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)") if dom["traffic"] else None,
                              "achievable_peak_measured": round(copy_gbs, 1),
                              "selection": "the bracketed kernel with the longest average duration in this run's own untimed sweep",
                              "lowest": lowest,
                              "per_kernel_ms": {n: round(ms, 4) for n, ms in sorted(ktimes.items(), key=lambda kv: -kv[1])},
                              "per_kernel_frac": {n: round(f, 4) for n, f in sorted(fracs.items(), key=lambda kv: kv[1])},
                              "whole_chain": {"achieved": round(chain_gbs, 1), "frac": round(chain_gbs / HBM_PEAK_GBS, 4), "traffic": chain_traffic,
                                              "algorithmic_bytes": round(chain_bpp * W * rows_gpu), "frac_of_achievable": round(chain_gbs / copy_gbs, 4)}}
        # per-stage sweep (separate frames, stage events of the chain; serial streams)
        if not args.no_pass_breakdown and not shared_frame:  # (the sharded mode steps all ranks together: no rank-0-only frames)
            passes = runner.time_passes(reps=10)
            result["roofline"]["per_pass_ms"] = {k: round(v["ms"], 4) for k, v in passes.items()}
            result["roofline"]["per_pass_frac"] = {k: round(v["algo_bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for k, v in passes.items() if v["algo_bytes"]}

    # ---------------------------------------------------------------- CPU baseline: the oracle / reference on the host cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(budget_s=20.0, size=(W, H), device=dev)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
            result["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}

    if shared_frame:
        # the sharded frame against the unsharded chain, bit for bit: every frame of the run with --verify-shard, else three frames after the timed region
        if args.verify_shard and runner.mifx_comm is None:
            n_cmp, n_bad = args.warmup + args.steps, runner.mismatches
        else:
            n_cmp = 3
            n_bad = runner.verify_against_unsharded(n_cmp)
        bad = torch.tensor([n_bad], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(bad)
        result["shard_verified"] = {"frames_compared": n_cmp, "bands_that_differed": int(bad.item()),
                                    "how": "every rank also runs the unsharded chain from the same history reset and compares its band of the output bit for bit"}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
