#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native DiligentFX hot path on synthetic G-buffers.

    python bench.py --gpus N --steps K --warmup W        (N > 1: one rank per GPU.  Started under torch.distributed.run -- WORLD_SIZE set -- the process is one of the
                                                          N ranks; started plainly it launches the N ranks itself: launch_ranks().  A world size that is not --gpus, or
                                                          fewer GPUs than ranks, is refused with an {"error": ...} line and a non-zero exit code, never timed on one GPU)

A "step" is one frame of the full chain (BASELINE.json configs[3]): PBR GGX+IBL shade -> PostFX prep -> SSR -> SSAO -> composite ->
TAA -> Bloom -> ToneMap at 3840x2160 per GPU, steady state (temporal history warmed up).  Inputs (the G-buffers of a pre-rendered camera
orbit of consecutive frames -- tiling.TiledChain.build_inputs -- and the IBL maps) are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

`--config ssao1080` / `--config pbr4k` run BASELINE configs[1] / [2] instead (SSAO alone on a 1920x1080 depth + normal G-buffer; the PBR shade alone at 3840x2160) and print the
same line for them: value, roofline of that configuration's dominant kernel, cpu_baseline of the checker on the same configuration.

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
# SSR resolved depth R16_FLOAT (2 B), closest motion RG16_FLOAT (4 B), Bloom levels R11G11B10_FLOAT (4 B), SSR's reflection mask one byte (round 3); depth, the depth
# pyramids and the motion input keep 4 / 8 bytes.  454.3 B/px for the whole chain (fp32 storage: 874.3).
ALGO_BPP_H4 = {"pbr_shade": 44.0, "prep": 24.0, "ssr": 166.67, "ssao": 80.0, "composite": 57.0, "taa": 36.0, "dof": 0.0, "bloom": 30.67, "tonemap": 16.0}
KERNEL_BPP_H4 = {"pbr_shade_kernel": 44.0, "pbr_shade_ssr_mask_kernel": 44.0 + 11.0, "bloom_upsample_tonemap_kernel": 17.0 + 16.0, "postfx_prep_kernel": 24.0,
                 "ssr_mask_roughness_kernel": 11.0, "ssr_intersection_kernel": 36.33, "ssr_spatial_kernel": 39.0, "ssr_temporal_kernel": 46.0, "ssr_bilateral_kernel": 29.0,
                 "ssao_compute_ao_kernel": 14.33, "ssao_temporal_kernel": 19.0, "ssao_resample_kernel": 17.67, "ssao_spatial_kernel": 17.0, "ssao_resolve_list_kernels": 17.67 + 17.0, "composite_kernel": 57.0, "composite_ssr_cleanup_kernel": 57.0 + 29.0,
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


_CORES = None  # (threads, description), taken once at start-up: after OpenMP has bound the main thread to its place the affinity mask reads as that place


def host_cores():
    global _CORES
    if _CORES is None:
        _CORES = usable_cores()
    return _CORES


def pin_host_threads():
    """OpenMP settings of the CPU baseline, set before any OpenMP runtime loads (libgomp reads them once): one thread per usable physical
    core, bound close.  Only when the caller has not chosen otherwise."""
    n, _ = host_cores()
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
    h4 = os.environ.get("MIFX_STORAGE") == "h4"
    cpu = cpu_chain.CpuChain(pyref.QuantizingLib(lib) if h4 else lib, pfx)  # --storage h4: the format-emulating checker (oracle/pyref.py)
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
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or host_cores()[0]
    return {"value": round(w * h * n / t_total / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": kind,
            "sample": f"{n} consecutive frame(s) of the full chain{' with every stored image rounded to its target format (numpy, single thread: part of the measured time)' if h4 else ''} at {w}x{h} after one warm-up frame, {t_total:.1f} s of CPU work "
                      f"({'oracle/_ref: reference shader source compiled for the CPU' if kind == 'reference' else 'oracle/mifx_oracle.cpp'}, OpenMP, "
                      f"OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')} OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}; host: {host_cores()[1]})"}


def valu_roof(w, h, ktimes):
    """Per kernel: vector instructions per pixel and the fraction of the VALU issue roof the kernel runs at, from the committed SQ counter pass of this build
    (profiles/r*_pmc_sq_counters*.txt: SQ_INSTS_VALU per dispatch) and the committed issue-rate measurement (profiles/r*_valu_issue_rate*.txt: ns per wave64
    v_fma_f32 per SIMD at 8 waves, measured over >= 50 ms with the shader clock recorded): frac = SQ_INSTS_VALU x issue_ns / (SIMDs x this run's kernel time).
    None when either file is missing or was taken at another resolution."""
    import glob
    import re

    sq = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_counters*.txt")))
    ir = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_issue_rate*.txt")))
    if not sq or not ir or (w, h) != (3840, 2160):
        return None
    issue = None
    for line in open(ir[-1]):
        m = re.match(r"v_fma_f32\s+W=8\s+.*?([0-9.]+) /\s*([0-9.]+) ns per wave-instruction", line)
        if m:
            issue = 0.5 * (float(m.group(1)) + float(m.group(2)))
            break
    if issue is None:
        return None
    head, rows = None, {}
    for line in open(sq[-1]):
        if line.startswith("kernel"):
            if head is not None:
                break  # (the file goes on with derived tables: the counters are the first one)
            head = line.split()
            continue
        m = re.match(r"^(mifx::.*?)\s+(\d+)\s+([0-9.]+)((?:\s+[0-9.]+|\s+nan)+)\s*$", line)  # name (may contain spaces), dispatches, duration, counters
        if m:
            rows[m.group(1).replace("mifx::", "").split("<")[0]] = [float(m.group(2)), float(m.group(3))] + [float(x) for x in m.group(4).split()]
    cols = [c for c in (head or []) if c.startswith("SQ_")]
    if "SQ_INSTS_VALU" not in cols:
        return None
    ci = cols.index("SQ_INSTS_VALU")
    alias = {"pbr_shade_ssr_mask_kernel": "pbr_shade_kernel", "bloom_upsample_tonemap_kernel": "bloom_final_tonemap_kernel", "composite_ssr_cleanup_kernel": "composite_kernel"}
    out = {"issue_ns": round(issue, 4), "issue_source": os.path.relpath(ir[-1], ROOT), "insts_source": os.path.relpath(sq[-1], ROOT), "simds": 1024, "per_kernel": {}}
    for name, ms in ktimes.items():
        r = rows.get(alias.get(name, name))
        if not r or ms <= 0:
            continue
        insts = r[2 + ci]  # after `disp` and `dur_us`
        out["per_kernel"][name] = {"insts_per_px": round(insts * 64.0 / (w * h), 1), "frac": round(insts * issue * 1e-9 / 1024.0 / (ms * 1e-3), 3)}
    return out if out["per_kernel"] else None


def read_pmc_table(path):
    """tools/pmc_stats.py output -> {kernel name without namespace and template arguments: {"disp": n, "dur_us": t, COUNTER: value per dispatch, ...}}."""
    import re

    head, rows = None, {}
    for line in open(path):
        if line.startswith("kernel"):
            if head is not None:
                break  # (derived tables follow)
            head = line.split()
            continue
        m = re.match(r"^(mifx::.*?)\s+(\d+)\s+([0-9.]+)((?:\s+[0-9.]+|\s+nan)+)\s*$", line)
        if m and head:
            vals = [float(x) for x in m.group(4).split()]
            name = m.group(1).replace("mifx::", "").split("<")[0]
            rows.setdefault(name, {"disp": float(m.group(2)), "dur_us": float(m.group(3))})
            for c, v in zip([c for c in head if c.startswith(("SQ_", "TCP_", "TCC_", "FETCH", "WRITE"))], vals):
                rows[name][c] = v
    return rows


# bracket name -> the instance of the kernel the default frame launches, as tools/isa_stats.py names it (static instruction mix: profiles/r*_isa_stats.txt)
ISA_KERNELS = {"pbr_shade_ssr_mask_kernel": "pbr_shade_kernel<false, false, true>", "pbr_shade_kernel": "pbr_shade_kernel<false, false, false>", "taa_kernel": "taa_kernel<false, true, false, false>",
               "ssr_intersection_kernel": "ssr_intersection_kernel<false, false, false>", "ssao_compute_ao_kernel": "ssao_compute_ao_kernel<0, false>", "composite_ssr_cleanup_kernel": "composite_kernel<0, true>",
               "ssr_spatial_kernel": "ssr_spatial_kernel<false>", "ssr_temporal_kernel": "mifx::ssr_temporal_kernel", "ssao_temporal_kernel": "ssao_temporal_kernel<true>",
               "bloom_upsample_tonemap_kernel": "bloom_final_tonemap_kernel<true, 4, true>", "bloom_prefilter_kernel": "bloom_prefilter_kernel<true>", "postfx_prep_kernel": "postfx_prep_kernel<false>"}


def valu_cost_factors():
    """Per kernel, the static issue cost of its vector instructions over their number (tools/isa_stats.py: full-rate opcodes 1, half-rate ones -- min / max / conversions /
    shifts / anything with an SGPR operand -- 2, transcendentals 4, from the measured table profiles/r03_valu_issue_rate.txt): what a count of instructions has to be
    multiplied with to become issue time."""
    import glob
    import re

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_stats.txt")))
    if not paths:
        return {}, None
    f = {}
    for line in open(paths[-1]):
        m = re.match(r"^(.*?)\s+total=\d+ valu=(\d+) .* cost=(\d+)", line)
        if m and int(m.group(2)) > 0:
            f[m.group(1).strip()] = float(m.group(3)) / float(m.group(2))
    return f, os.path.relpath(paths[-1], ROOT)


def speed_of_light(w, h, ktimes, copy_gbs, clock_ghz=2.4, cus=256, simds=1024):
    """Per bracketed kernel, what each of the three resources it competes for would take alone -- from the committed counter passes of this build -- beside what it takes:
       hbm_us  = the kernel's measured HBM bytes (FETCH_SIZE / WRITE_SIZE passes) at the copy rate this run measured
       tcp_us  = its vector-L1 tag look-ups (TCP_TOTAL_CACHE_ACCESSES: one per clock and CU, tools/microbench/tcp_gather_rate.hip) / (CUs x clock)
       valu_us = its vector instructions (SQ_INSTS_VALU) x the measured full-rate issue cost / SIMDs        (a LOWER estimate: half-rate and transcendental opcodes cost 2 - 4x)
       valu_weighted_us = valu_us x the kernel's static cost per instruction (valu_cost_factors: 1.3 - 1.6): the issue time of the instructions it actually consists of
       actual_us = this run's own one-stream duration;  floor_us = the largest of HBM, L1 and weighted VALU;  residual_us = actual - floor (latency the resident waves do
       not cover, and whatever part of the three does not overlap).  None when the counter files of this resolution are not committed."""
    import glob

    tr = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    tcp = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_tcp_*.txt")))
    valu = valu_roof(w, h, ktimes)
    if not tr or not tcp or (w, h) != (3840, 2160):
        return None
    t = json.load(open(tr[-1]))
    if t["resolution"] != [w, h]:
        return None
    tcp_rows = read_pmc_table(tcp[-1])
    factors, factor_src = valu_cost_factors()
    alias = {"pbr_shade_ssr_mask_kernel": ["pbr_shade_kernel"], "bloom_upsample_tonemap_kernel": ["bloom_final_tonemap_kernel"], "composite_ssr_cleanup_kernel": ["composite_kernel"],
             "ssao_resolve_list_kernels": ["ssao_resample_list_kernel", "ssao_spatial_list_kernel"]}
    table, tot = {}, {"actual_us": 0.0, "floor_us": 0.0, "hbm_us": 0.0, "tcp_us": 0.0, "valu_us": 0.0, "valu_weighted_us": 0.0}
    for name, ms in sorted(ktimes.items(), key=lambda kv: -kv[1]):
        parts = alias.get(name, [name])
        got = [v for k, v in t["kernels"].items() if any(k.startswith(p_) for p_ in parts)]
        bytes_ = sum(v["read_bytes"] + v["write_bytes"] for v in got) if got else None
        acc = [tcp_rows[p_]["TCP_TOTAL_CACHE_ACCESSES_sum"] for p_ in parts if p_ in tcp_rows and "TCP_TOTAL_CACHE_ACCESSES_sum" in tcp_rows[p_]]
        row = {"actual_us": round(ms * 1e3, 1),
               "hbm_us": round(bytes_ / (copy_gbs * 1e9) * 1e6, 1) if bytes_ else None,
               "tcp_us": round(sum(acc) / (cus * clock_ghz * 1e9) * 1e6, 1) if len(acc) == len(parts) else None,
               "valu_us": round(valu["per_kernel"][name]["frac"] * ms * 1e3, 1) if valu and name in valu["per_kernel"] else None}
        want = ISA_KERNELS.get(name, "")
        fac = factors.get(want) or next((v for k, v in factors.items() if want and k.startswith(want.rstrip(">"))), None)  # (a template argument added later keeps the prefix)
        row["valu_weighted_us"] = round(row["valu_us"] * fac, 1) if (row["valu_us"] is not None and fac) else None
        known = [v for v in (row["hbm_us"], row["tcp_us"], row["valu_weighted_us"] if row["valu_weighted_us"] is not None else row["valu_us"]) if v is not None]
        if known:
            row["floor_us"] = max(known)
            row["residual_us"] = round(row["actual_us"] - row["floor_us"], 1)
            tot["actual_us"] += row["actual_us"]
            tot["floor_us"] += row["floor_us"]
            for k in ("hbm_us", "tcp_us", "valu_us", "valu_weighted_us"):
                tot[k] += row[k] or 0.0
        table[name] = row
    return {"per_kernel": table, "bracketed_kernels_total": {k: round(v, 1) for k, v in tot.items()},
            "whole_frame_hbm_us": round(t["chain_traffic"] / (copy_gbs * 1e9) * 1e6, 1),
            "sources": {"bytes": os.path.relpath(tr[-1], ROOT), "tcp": os.path.relpath(tcp[-1], ROOT), "valu": valu["insts_source"] if valu else None, "valu_weights": factor_src, "copy_rate_gbs": round(copy_gbs, 1),
                        "clock_ghz": clock_ghz},
            "note": "actual = this run's untimed one-stream sweep; the counters are the committed passes of the same build (they cannot be read inside a timed run); kernels without "
                    "a bracket (pyramids, Bloom levels: ~0.2 ms of small launches) are not in the table"}


def traffic_file():
    """(the latest committed PMC traffic file of this storage build as a dict, its path, whether its counters were taken from the device code this process has loaded)."""
    import glob

    suffix = "_h4" if os.environ.get("MIFX_STORAGE") == "h4" else ""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{suffix}.json")))  # the latest round's measurement of this storage build
    if not paths:
        return None, None, None
    t = json.load(open(paths[-1]))
    try:
        from diligentfx_amd import binding as B

        loaded = B.device_code_sha16()
    except Exception:  # noqa: BLE001
        loaded = None
    same = bool(loaded) and t.get("device_code_sha16") == loaded
    return t, os.path.relpath(paths[-1], ROOT), {"matches": same, "loaded": loaded, "profiled": t.get("device_code_sha16")}


def pmc_traffic(w, h, kernel):
    """HBM bytes per launch of `kernel` (and per frame of the whole chain) from the committed PMC passes -- counters cannot be read inside a
    timed run; (None, None, None) when no measurement exists for this resolution OR the measurement belongs to other kernels than the ones loaded (the file carries the
    SHA-256 of the profiled library's device code: traffic_file)."""
    t, path, ident = traffic_file()
    if t is None or t["resolution"] != [w, h] or not ident["matches"]:
        return None, None, None
    k = next((v for name, v in t["kernels"].items() if name.startswith(kernel)), None)
    return (k["read_bytes"] + k["write_bytes"]) if k else None, t["chain_traffic"], path


# bracket name (KERNEL_BPP) -> the kernels of the PMC file it covers (prefixes of the demangled names)
PMC_KERNELS = {"pbr_shade_ssr_mask_kernel": ["pbr_shade_kernel"], "pbr_shade_kernel": ["pbr_shade_kernel"], "composite_ssr_cleanup_kernel": ["composite_kernel"], "composite_kernel": ["composite_kernel"],
               "bloom_upsample_tonemap_kernel": ["bloom_final_tonemap_kernel"], "ssao_resolve_list_kernels": ["ssao_resample_list_kernel", "ssao_spatial_list_kernel"]}


def measured_byte_fractions(w, h, ktimes):
    """Per bracketed kernel: the HBM bytes the committed PMC passes measured for it over this run's own duration, as a fraction of the 8 TB/s roof -- beside
    per_kernel_frac, which credits a fused kernel the algorithmic bytes of every reference pass it performs (the composite with R7 inside is credited planes it no longer
    moves and reads > 1 there).  None when no measurement of this resolution is committed."""
    import glob

    suffix = "_h4" if os.environ.get("MIFX_STORAGE") == "h4" else ""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{suffix}.json")))
    if not paths:
        return None
    t = json.load(open(paths[-1]))
    if t["resolution"] != [w, h]:
        return None
    out = {}
    for name, ms in ktimes.items():
        prefixes = PMC_KERNELS.get(name, [name])
        got = [v for k, v in t["kernels"].items() if any(k.startswith(p) for p in prefixes)]
        if got and ms > 0:
            out[name] = round(sum(v["read_bytes"] + v["write_bytes"] for v in got) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return {"source": os.path.relpath(paths[-1], ROOT), "build": t.get("build"), "frac": dict(sorted(out.items(), key=lambda kv: kv[1]))}


def measured_copy_peak(runner, dev, torch):
    """Achievable HBM rate of this device (SURVEY 8d asks for it beside the 8 TB/s spec): a 1 GiB device-to-device copy with the library's own streaming access
    pattern (mifx_debug_stream_copy: one 16-byte texel per lane, as the chain's streaming passes), read + write bytes; median of 10 copies."""
    import ctypes

    from diligentfx_amd import binding as B

    n = 1 << 28
    a, b = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    a.fill_(1.0)
    lib, ctx = runner.chain.lib, runner.chain.postfx
    ctx.sync_stream()
    copy = lambda: B.check(lib.mifx_debug_stream_copy(ctx.handle, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_uint64(4 * n)))  # noqa: E731
    for _ in range(3):
        copy()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for i in range(10):
        copy()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))
    return 2.0 * 4.0 * n / (0.5 * (ms[4] + ms[5]) * 1e-3) / 1e9


def warm_up(step, torch, frames, seconds=0.3):
    """At least `frames` calls of `step` AND `seconds` of device work before a short measurement: after the host-side phase in front of a stage line (building inputs, the
    CPU baseline) the device needs tens of milliseconds of work to reach its sustained clock -- the PBR shade alone measured 208 us per launch in a 12 + 40 frame run started
    cold and 181 us in the same run repeated (profiles/r06_exp_shade_clock.txt); round 5's stage lines were taken on that ramp."""
    t0, n = time.perf_counter(), 0
    while n < frames or time.perf_counter() - t0 < seconds:
        for _ in range(10):
            step()
        n += 10
        torch.cuda.synchronize()
    return n


def tonemap_line(device_index, tables, torch, with_cpu, steps=200, warmup=20, size=(1920, 1080)):
    """BASELINE configs[0]: ToneMapping only on a 1920x1080 synthetic HDR float4 buffer (ToneMap(), Shaders/PostProcess/ToneMapping/public/ToneMapping.fxh:87-226) --
    tonemap_kernel on its own, 16 B/px read + 16 written, for the bench's operator (Uncharted2 + sRGB) and for AgX; beside it, with_cpu, the reference's own shader source
    (oracle/_ref; the hand port where that did not travel) on the host cores: the configuration's "CPU reference path"."""
    import numpy as np

    from diligentfx_amd import api, binding as B, synth

    w, h = size
    ctx = api.PostFXContext(device_index, tables["sobol_256d"], tables["scrambling_tile"])
    hdr = synth.make_hdr_buffer(w, h, ctx.device)
    ldr = torch.empty_like(hdr)
    out = {"workload": f"ToneMap() on a {w}x{h} synthetic HDR float4 buffer (BASELINE configs[0]); 16 B/px read + 16 written", "algorithmic_bytes_per_px": 32.0, "modes": {}}
    for name, mode, flags in (("uncharted2_srgb", 4, 1), ("agx", 8, 0)):
        attr = B.ToneMappingAttribs.default(mode)
        warm_up(lambda: ctx.tone_map(hdr, attr, 0.3, flags, out=ldr), torch, warmup, seconds=0.15)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(steps):
            ctx.tone_map(hdr, attr, 0.3, flags, out=ldr)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / steps
        gbs = 32.0 * w * h / (ms * 1e-3) / 1e9
        line = {"ms_per_step": round(ms, 5), "value": round(w * h / (ms * 1e-3) / 1e6, 1), "unit": "Mpixels/s", "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "steps": steps}
        if with_cpu:
            try:
                for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
                    if p not in sys.path:
                        sys.path.insert(0, p)
                import pyref

                lib, pfx, kind = pyref.ref_lib(), "ref_", "reference"
                if lib is None:
                    lib, pfx, kind = pyref.oracle_lib(), "oracle_", "port"
                src, dst = hdr.cpu().numpy(), np.zeros((h, w, 4), np.float32)
                lib.call(pfx + "tonemap", [src], [dst], attribs=bytes(attr), fval=[0.3], ival=[flags])  # (page-in)
                t0, n = time.perf_counter(), 0
                while n < 3 or (time.perf_counter() - t0 < 1.0 and n < 50):
                    lib.call(pfx + "tonemap", [src], [dst], attribs=bytes(attr), fval=[0.3], ival=[flags])
                    n += 1
                cpu_ms = (time.perf_counter() - t0) / n * 1e3
                line.update({"cpu_ms": round(cpu_ms, 3), "cpu_value": round(w * h / (cpu_ms * 1e-3) / 1e6, 1), "cpu_kind": kind, "cpu_cores": int(os.environ.get("OMP_NUM_THREADS", "0")) or host_cores()[0],
                             "cpu_sample": f"{n} frames after one warm-up"})
            except Exception as e:
                line["cpu_failed"] = repr(e)
        out["modes"][name] = line
    out["note"] = "1920x1080 x 32 B = 66 MB per launch: the source and the target fit the 256 MB Infinity Cache, so back-to-back launches are partly served by it (frac may exceed what HBM alone gives)"
    ctx.close()
    return out


def stage_lines(device_index, tables, torch, steps=40, warmup=12, with_cpu=True, layers=False):
    """BASELINE configs[0] (ToneMapping only, 1920x1080: tonemap_line), configs[1] (PostFX prep + SSAO on a 1920x1080 depth + normal G-buffer) and configs[2] (the PBR GGX + IBL
    shade at 3840x2160) on their own, one stream, K frames bracketed by events after a warm-up: ms per frame, Mpixels/s and the fraction of the 8 TB/s roof their
    algorithmic bytes (32, 176 and 84 per pixel) reach.  The same measurement as `--config ssao1080 | pbr4k`, shortened so that the default command carries it."""
    from diligentfx_amd import tiling

    out = {}
    try:
        out["tonemap1080"] = tonemap_line(device_index, tables, torch, with_cpu)
    except Exception as e:
        out["tonemap1080"] = {"failed": repr(e)}
    for key, mode, (w, h), bpp in (("ssao1080", "ssao", (1920, 1080), ALGO_BPP["prep"] + ALGO_BPP["ssao"]), ("pbr4k", "pbr", (3840, 2160), ALGO_BPP["pbr_shade"])):
        r = tiling.StageRunner(mode, device_index, tables["sobol_256d"], tables["scrambling_tile"], w, h)
        r.build_inputs(n_frames=8)
        warmed = warm_up(r.step, torch, warmup)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(steps):
            r.step()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / steps
        gbs = bpp * w * h / (ms * 1e-3) / 1e9
        out[key] = {"workload": "PostFX prep + SSAO A2..A8, 1920x1080 (BASELINE configs[1])" if mode == "ssao" else "PBR GGX + IBL shade, 3840x2160 (BASELINE configs[2])",
                    "ms_per_step": round(ms, 4), "value": round(w * h / (ms * 1e-3) / 1e6, 1), "unit": "Mpixels/s", "algorithmic_bytes_per_px": round(bpp, 1),
                    "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "steps": steps, "warmup": warmed,
                    "warmup_note": "frames until 0.3 s of device work had run: a stage measured right behind a host-side phase is on the clock ramp otherwise (profiles/r06_exp_shade_clock.txt)"}
        del r
        torch.cuda.empty_cache()
    if layers:  # --layers-line: the shade with all five material layers (mifx_pbr_shade_execute_layers) -- outside SURVEY section 8, not in the default line
        try:
            out["pbr4k_layers"] = layered_shade_line(device_index, tables, torch, steps=steps // 2, warmup=warmup // 2)
        except Exception as e:
            out["pbr4k_layers"] = {"failed": repr(e)}
    return out


def layered_shade_line(device_index, tables, torch, steps, warmup, size=(3840, 2160)):
    from diligentfx_amd import api, tiling

    w, h = size
    r = tiling.StageRunner("pbr", device_index, tables["sobol_256d"], tables["scrambling_tile"], w, h)
    r.build_inputs(n_frames=2)
    g = r._frame_view(*r.orbit_position(0))
    dev = g["depth"].device
    gen = torch.Generator(device=dev).manual_seed(5)
    u = lambda *s: torch.rand(*s, device=dev, generator=gen)  # noqa: E731
    n = g["normal"][..., :3].float()
    t = u(h, w, 3) - 0.5
    t = torch.nn.functional.normalize(t - n * (t * n).sum(-1, keepdim=True), dim=-1)
    z = torch.zeros(h, w, 1, device=dev)
    ang = 6.2831853 * u(h, w, 1)
    planes = {"clearcoat": torch.cat([u(h, w, 1), 0.05 + 0.95 * u(h, w, 1), z, z], -1), "clearcoat_normal": torch.cat([torch.nn.functional.normalize(n + 0.3 * (u(h, w, 3) - 0.5), dim=-1), z], -1),
              "sheen": torch.cat([u(h, w, 3), 0.05 + 0.95 * u(h, w, 1)], -1), "anisotropy": torch.cat([torch.cos(ang), torch.sin(ang), u(h, w, 1), z], -1), "tangent": torch.cat([t, z], -1),
              "iridescence": torch.cat([u(h, w, 1), 100.0 + 300.0 * u(h, w, 1), z, z], -1), "transmission": u(h, w), "sheen_albedo_scaling_lut": 0.5 * u(32, 32),
              "preintegrated_charlie": 0.3 * u(32, 32)}
    planes = {k: v.contiguous() for k, v in planes.items()}
    gb = {k: g[k] for k in ("base_color", "normal", "material", "depth")}
    step = lambda: api.pbr_shade_layers(r.ctx, gb, planes, 31, g["camera"], r.shade, r.ibl, iridescence_ior=1.33, anisotropy_rotation=0.7)  # noqa: E731
    for _ in range(max(warmup, 2)):
        step()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(steps):
        step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    bpp = 84.0 + 100.0  # the default shade's planes + clear coat and its normal, sheen, anisotropy and the tangent, iridescence (16 B each), transmission (4 B)
    gbs = bpp * w * h / (ms * 1e-3) / 1e9
    return {"workload": "PBR GGX + IBL shade with clear coat + sheen + anisotropy + iridescence + transmission, 3840x2160 (DESIGN.md 6b; per-layer figures: profiles/r04_layers_timing_v3.txt)",
            "ms_per_step": round(ms, 4), "value": round(w * h / (ms * 1e-3) / 1e6, 1), "unit": "Mpixels/s", "algorithmic_bytes_per_px": bpp, "achieved": round(gbs, 1),
            "frac": round(gbs / HBM_PEAK_GBS, 4), "steps": steps, "warmup": max(warmup, 2)}


METRIC_CHAIN = "Mpixels/s full PBR+postFX chain @4K; %HBM roofline; 1/2/4/8-GPU scaling"


def error_line(args, message, **extra):
    """The one JSON line of a run that must not be mistaken for a measurement."""
    line = {"metric": METRIC_CHAIN, "error": message, "value": None, "unit": "Mpixels/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True}
    line.update(extra)
    print(json.dumps(line), flush=True)


def rank_environment():
    """(world, rank, local rank) when a torch.distributed launcher started this process, None otherwise."""
    if "WORLD_SIZE" not in os.environ:
        return None
    return int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process becomes the launcher -- N ranks of this same file under torch.distributed.run
    (one per GPU, rendezvous on 127.0.0.1 at a free port), their output passed through.  Returns the exit code; prints an error line itself when the ranks end without a
    result line, so that the caller always gets exactly one line to parse."""
    import socket
    import subprocess

    if not args.single_gpu and not args.dry_run_ranks:
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            error_line(args, f"--gpus {args.gpus}: this node shows {have} GPU(s); refusing to time fewer GPUs than asked for (--single-gpu puts every rank on cuda:0, for tests)", gpus_visible=have)
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL and cross-process device memory need on this driver
    env["MIFX_BENCH_LAUNCHER"] = "self"
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    got = False
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        got = got or (line.startswith("{") and '"metric"' in line)
    rc = proc.wait()
    if not got:
        error_line(args, f"the {args.gpus} ranks ended with exit code {rc} and no result line (their stderr is above)", launcher_cmd=" ".join(cmd))
        return rc or 3
    return rc


def parse_args(argv=None):
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
    p.add_argument("--backend", default=None, help="torch.distributed backend of the SIDE CHANNEL (the unique id, band times, flags, the closing barrier).  Default: gloo with "
                   "--comm rccl -- the frame's exchanges run inside libmifx on its own RCCL communicator, so no second (torch) RCCL communicator is created; nccl with --comm torch, "
                   "where torch.distributed moves the rows.  gloo with --single-gpu exercises the multi-rank code on one GPU")
    p.add_argument("--dry-run-hang-rank", type=int, default=-1, help="testing, with --dry-run-ranks: this rank never finishes (the watchdog's case)")
    p.add_argument("--dry-run-ranks", action="store_true", help="testing (no GPU needed): every rank joins the process group, the ranks agree on the world size and rank 0 prints "
                   "a line with n_gpus and the ranks seen -- the launch plumbing of `python bench.py --gpus N` without the frames")
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
    p.add_argument("--config", default="chain", choices=("chain", "ssao1080", "pbr4k"), help="chain: the full chain (BASELINE configs[3]; N > 1: configs[4]) -- the headline; ssao1080: "
                   "configs[1], PostFX prep + SSAO on a 1920x1080 depth + normal G-buffer; pbr4k: configs[2], the PBR GGX + IBL shade alone at 3840x2160")
    p.add_argument("--overlap", type=int, default=None, choices=(0, 1, 2, 3, 4, 5), help="mifx_chain_set_overlap: 1 = prep + SSAO on a second stream, 2 = also across frames (inputs resident), "
                   "3 = three lanes across frames (shade + prep + Hi-Z + SSAO | SSR + composite + TAA | Bloom), 4 = those lanes with two frames in flight, 5 = 4 with the composite + TAA on the Bloom lane (the default for N = 1)")
    p.add_argument("--lane-edges", default=None, help="mode 4: mifx_chain_set_lane_edges (\"waiter<signal@frames,...\")")
    p.add_argument("--fusion-mask", type=lambda v: int(v, 0), default=None, help="A/B: mifx_chain_set_fusion_mask (MIFX_CHAIN_FUSE_*; default: every fusion on; 3 = round 2's chain)")
    p.add_argument("--no-calibrate", action="store_true", help="N > 1, one shared frame: keep the band heights of the three-class cost model instead of refining them from measured band times "
                   "before the warm-up (TiledChain.calibrate_cuts)")
    p.add_argument("--cuts", type=lambda v: [int(x) for x in v.split(",")], default=None, help="N > 1, one shared frame: the row cuts (N + 1 values from 0 to the frame height, e.g. a "
                   "line's config.band_calibration.final_cuts) instead of the cost model's -- replays a recorded run; implies --no-calibrate")
    p.add_argument("--watchdog-s", type=float, default=900.0, help="N > 1: a rank that has not finished after this many seconds prints an error line (rank 0: the JSON line) and "
                   "exits with code 4, which makes the launcher end the others -- an exchange whose peer never answers must not hang the node; 0 = off")
    p.add_argument("--exact-warmup", action="store_true", help="profiling runs (rocprofv3 counts frames): exactly --warmup frames in front of the timed region; by default N = 1 keeps "
                   "warming up until 0.3 s of device work have run (warm_up)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-overlap-check", action="store_true", help="skip overlap_verified (the run's stream mode against the one-stream chain, bit for bit, after the timed region)")
    p.add_argument("--no-single-gpu-reference", action="store_true", help="N > 1: skip single_gpu_same_frame_ms (the whole shared frame on rank 0's GPU alone, after the timed region)")
    p.add_argument("--no-stage-lines", action="store_true", help="skip config.stage_lines (BASELINE configs[0], [1] and [2] measured after the timed region)")
    p.add_argument("--layers-line", action="store_true", help="also time the PBR shade with all five material layers (config.stage_lines.pbr4k_layers; out of SURVEY section 8's scope)")
    p.add_argument("--no-pass-breakdown", action="store_true")
    p.add_argument("--no-kernel-sweep", action="store_true", help="profiling runs (rocprofv3 counts frames): skip the untimed per-kernel sweep; the line then carries no `roofline`")
    return p.parse_args(argv)


def cpu_baseline_stage(mode, size, device, budget_s=15.0):
    """The checker on the host cores for --config ssao1080 / pbr4k: consecutive frames of the same orbit at the configuration's own size, one warm-up frame, then
    frames until ~budget_s of CPU work (inputs rendered outside the timed region)."""
    import numpy as np
    import torch

    from diligentfx_amd import binding as B, synth

    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import chain_util
    import cpu_chain
    import pyref
    from util import blue_noise_tables

    lib, pfx, kind = pyref.ref_lib(), "ref_", "reference"
    if lib is None:
        lib, pfx, kind = pyref.oracle_lib(), "oracle_", "port"
    w, h = size
    cpu = cpu_chain.CpuChain(lib, pfx)
    scene = synth.Scene()
    ibl = chain_util.make_ibl(lib, pfx, env_size=64, lut_size=64, irr_size=16, pref_size=32, lut_samples=64, irr_samples=128, pref_samples=32) if mode == "pbr" else None
    sa = chain_util.shade_attribs(len(ibl["prefiltered"]) - 1) if ibl else None
    tables = blue_noise_tables()
    t_total, n = 0.0, 0
    for i, fr in enumerate(range(16, 16 + 64)):
        f = synth.make_frame(scene, fr, w, h, device if device is not None else torch.device("cpu"))
        g = {k: v.cpu().numpy() for k, v in f.items() if isinstance(v, torch.Tensor)}
        cam, prev = bytes(f["camera"]), bytes(f["prev_camera"])
        t0 = time.perf_counter()
        if mode == "ssao":
            pf = cpu.postfx(fr, g["depth"], g["prev_depth"], g["motion"], cam, prev, tables)
            cpu.ssao(pf, g["depth"], g["normal"], B.SSAOAttribs.default())
        else:
            rad, spec = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
            cpu.call("pbr_shade", [g["base_color"], g["normal"], g["material"], g["depth"], None, None, ibl["lut"], ibl["irradiance"], ibl["prefiltered"]], [rad, spec], cam0=cam,
                     attribs=bytes(sa), fval=[0.02, 0.03, 0.05, 0.0])
        dt = time.perf_counter() - t0
        if i > 0:  # (frame 0: page-in, history reset)
            t_total += dt
            n += 1
        if t_total > budget_s:
            break
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or host_cores()[0]
    what = "PostFX prep + SSAO (A2..A8, temporal history)" if mode == "ssao" else "the PBR shade (GGX + IBL)"
    return {"value": round(w * h * n / t_total / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": kind,
            "sample": f"{n} consecutive frame(s) of {what} at {w}x{h} after one warm-up frame, {t_total:.1f} s of CPU work "
                      f"({'oracle/_ref: reference shader source compiled for the CPU' if kind == 'reference' else 'oracle/mifx_oracle.cpp'}, OpenMP, "
                      f"OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')} OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}; host: {host_cores()[1]})"}


def kernel_sweep(runner, names, frames=3):
    """Untimed: every bracketed kernel in turn is timed with HIP events over `frames` consecutive frames -> {kernel: average ms per launch}."""
    times = {}
    for name in names:
        runner.arm_kernel_timing(name, frames)
        for _ in range(frames):
            runner.step()
        kt = runner.kernel_times_ms(frames)
        if kt:
            times[name] = sum(kt) / len(kt)
    runner.arm_kernel_timing(None, 0)
    return times


def stage_bytes(algo_bpp, kernel_bpp, fusion_mask):
    """Algorithmic bytes per pixel of the chain's profiled stages with the fused passes counted where they run: R2 inside the shade, R7 inside the composite, the
    copy-frame ToneMap inside Bloom's final pass (the stage it left keeps no bytes and is dropped from the per-stage fractions)."""
    b = dict(algo_bpp)
    if fusion_mask & 2:  # MIFX_CHAIN_FUSE_SSR_MASK_INTO_SHADE
        r2 = kernel_bpp["ssr_mask_roughness_kernel"]
        b["ssr"] -= r2
        b["pbr_shade"] += r2
    if fusion_mask & 4:  # MIFX_CHAIN_FUSE_SSR_CLEANUP_INTO_COMPOSITE
        r7 = kernel_bpp["ssr_bilateral_kernel"]
        b["ssr"] -= r7
        b["composite"] += r7
    if fusion_mask & 1:  # MIFX_CHAIN_FUSE_TONE_MAP_INTO_BLOOM
        b["bloom"] += b["tonemap"]
        b["tonemap"] = 0.0
    return b


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    env = rank_environment()
    if env is None and args.gpus > 1:
        return launch_ranks(args, argv)  # (plain `python bench.py --gpus N`: this process is the launcher of the N ranks)
    world, rank, local_rank = env if env is not None else (1, 0, 0)
    if world != args.gpus:
        # never time a world that is not the one asked for (round 5 printed a one-GPU line with n_gpus 1 for `--gpus 8` outside a launcher)
        if rank == 0:
            error_line(args, f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE={os.environ.get('WORLD_SIZE')}): refusing to run", world_size=world)
        return 2
    host_cores()  # (before anything binds this thread)
    global KERNEL_BPP, ALGO_BPP, CHAIN_BPP
    if args.storage == "h4":
        os.environ["MIFX_STORAGE"] = "h4"  # read when diligentfx_amd.binding is imported
        KERNEL_BPP, ALGO_BPP = KERNEL_BPP_H4, ALGO_BPP_H4
        CHAIN_BPP = sum(ALGO_BPP.values())
        # (the CPU baseline of this configuration is the checker with format emulation: every image a reference pass writes is rounded to its target format on the host)
    if world == 1 and not args.no_cpu_baseline:
        pin_host_threads()  # before any OpenMP runtime is loaded
    import numpy as np
    import torch
    import torch.distributed as dist

    if args.single_gpu:
        local_rank = 0
    backend = args.backend or ("gloo" if (args.comm == "rccl" or args.dry_run_ranks) else "nccl")
    args.backend = backend
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the side channel never leaves it (and the container's hostname may not resolve)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    watchdog = None
    if world > 1 and args.watchdog_s > 0:
        import threading

        def give_up():
            msg = f"watchdog: rank {rank} of {world} was still running after {args.watchdog_s:.0f} s (an exchange without an answer, or a rank that died?)"
            if rank == 0:
                error_line(args, msg)
            sys.stderr.write(msg + "\n")
            sys.stderr.flush()
            os._exit(4)

        watchdog = threading.Timer(args.watchdog_s, give_up)
        watchdog.daemon = True
        watchdog.start()
    if args.dry_run_ranks:  # the launch plumbing alone (tests/test_bench_host.py)
        seen = [None] * world
        if world > 1:
            dist.all_gather_object(seen, (rank, local_rank, os.getpid()))
            dist.barrier()
        else:
            seen = [(rank, local_rank, os.getpid())]
        if rank == 0:
            print(json.dumps({"metric": METRIC_CHAIN, "dry_run": True, "value": None, "n_gpus": world, "ranks": [list(x) for x in seen], "backend": backend,
                              "launcher": os.environ.get("MIFX_BENCH_LAUNCHER", "external")}), flush=True)
        if args.dry_run_hang_rank == rank:  # (test of the watchdog: this rank never finishes)
            time.sleep(3600)
        if watchdog is not None:
            watchdog.cancel()
        if world > 1:
            dist.destroy_process_group()
        return 0
    if not args.single_gpu and torch.cuda.device_count() < world:
        if rank == 0:
            error_line(args, f"{world} ranks but {torch.cuda.device_count()} GPU(s) visible: refusing to put two ranks on one GPU (--single-gpu does that, for tests)")
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    side_dev = dev if backend == "nccl" else torch.device("cpu")  # where the side channel's tensors live

    from diligentfx_amd import api, binding as B, synth
    from diligentfx_amd import tiling

    tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
    W, H = args.width, args.height
    stage = args.config != "chain"
    if stage:
        assert world == 1 and args.storage == "fp32", "--config ssao1080 / pbr4k: one GPU, fp32 planes"
        if args.config == "ssao1080" and (args.width, args.height) == (3840, 2160):
            W, H = 1920, 1080
    # N > 1: ONE frame of twice the width and height of the per-GPU configuration (7680x4320 = BASELINE configs[4]) sharded by row bands over
    # the ranks -- north_star's tile-parallel frame; --replicas keeps the N independent views of round 1
    shard = world > 1 and not args.replicas
    if shard and not args.shard_rows and (args.width, args.height) == (3840, 2160):
        W, H = 2 * args.width, 2 * args.height
    if shard and args.orbit_frames == 24:
        args.orbit_frames = 8  # 2.3 GB per resident 8K frame and rank

    # ---------------------------------------------------------------- inputs (resident in HBM before timing)
    if stage:
        runner = tiling.StageRunner("ssao" if args.config == "ssao1080" else "pbr", local_rank, tables["sobol_256d"], tables["scrambling_tile"], W, H)
    else:
        runner = tiling.TiledChain(local_rank, tables["sobol_256d"], tables["scrambling_tile"], rank, world, W, H, shard_rows=shard, verify=args.verify_shard, comm_backend=args.comm, cuts=args.cuts,
                                    fallback_backend=None if args.single_gpu else "nccl")
    shared_frame = runner.shard_rows
    runner.build_inputs(n_frames=args.orbit_frames)
    fusion_mask = 31 if args.fusion_mask is None else args.fusion_mask
    if args.fusion_mask is not None:
        runner.chain.set_fusion_mask(args.fusion_mask)
    # the chain on one GPU runs as three lanes sliding across frames (mifx_chain_set_overlap 3 / 4: the inputs of every frame are resident before the timed region, which
    # is those modes' contract; round 3 ran mode 2, measured 0.5 - 3.4 % slower on five boxes in round 4); --overlap 0 gives the serial chain, whose kernel durations are attributable
    # (second session of round 6: mode 4 -- the same three lanes with two frames in flight -- is the default.  Round 5 measured it equal to mode 3 (1.6925 against 1.6975 ms); with
    #  the streaming passes walking towards what the Infinity Cache holds and the dead-end traffic non-temporal it is 2.1 % faster on two boxes: profiles/r06_ab_rows_up.txt)
    # (... and mode 5 -- mode 4 with the composite, TAA and depth of field on the Bloom lane, so that the next frame's ray march runs beside them -- another 1.0 - 1.2 % on two boxes)
    overlap = args.overlap if args.overlap is not None else (5 if not stage and not shared_frame else 0)
    if overlap and not shared_frame:
        runner.chain.set_overlap(overlap)
        if args.lane_edges is not None:
            runner.chain.set_lane_edges(args.lane_edges)
    stage_bpp = stage_bytes(ALGO_BPP, KERNEL_BPP, fusion_mask)
    tiling.ALGO_BPP.update(stage_bpp)
    chain_bpp = CHAIN_BPP
    kernels = dict(KERNEL_BPP)
    if args.config == "ssao1080":
        chain_bpp = ALGO_BPP["prep"] + ALGO_BPP["ssao"]
        kernels = {k: v for k, v in KERNEL_BPP.items() if k.startswith(("ssao_", "postfx_prep"))}
    elif args.config == "pbr4k":
        chain_bpp = ALGO_BPP["pbr_shade"]
        kernels = {"pbr_shade_kernel": KERNEL_BPP["pbr_shade_kernel"]}
    if args.ssao_half or args.ssr_half:
        assert not stage, "--ssao-half / --ssr-half: options of the chain"
        runner.apply_option(lambda c: c.set_effect_feature_flags(ssao_feature_flags=2 if args.ssao_half else 0, ssr_feature_flags=2 if args.ssr_half else 0))
    if args.dof:
        assert not stage, "--dof: an option of the chain"
        for f in runner.frames:
            f["camera"].fFocusDistance, f["camera"].fFStop, f["camera"].fFocalLength = DOF_LENS
        runner.apply_option(lambda c: c.set_depth_of_field(B.DOFAttribs.default(), api.DepthOfField.FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING))
        tiling.ALGO_BPP["dof"] = DOF_BPP
        chain_bpp += DOF_BPP

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one shared frame: band heights fed back from measured band times (two rounds: every rank times its own band without the exchanges, the times are all-gathered, the
    # cuts move towards equal times, the histories start again) -- before the warm-up, so that the timed region runs on settled bands with a settled history
    calibration = None
    if shared_frame and not args.no_calibrate and not args.verify_shard and args.cuts is None:
        calibration = runner.calibrate_cuts(rounds=2, frames=6)
    warmup_frames = args.warmup
    if world == 1 and not args.exact_warmup:
        # at least --warmup frames, and at least 0.3 s of device work: the inputs were just rendered and the IBL maps precomputed -- seconds of host-side work during which the
        # device's clock has dropped (warm_up; the stage configurations are a few hundred microseconds per frame, five of them are over before the clock is back)
        warmup_frames = warm_up(runner.step, torch, args.warmup)
    else:  # (every rank must step the same number of frames: the exchanges live inside step(); the band calibration in front of this loop is ~100 frames of work)
        for i in range(args.warmup):
            runner.step()
    # which kernel is the frame's longest, and which is furthest below its roofline: measured here, not assumed (every rank steps the same
    # number of frames: the sharded mode exchanges data inside step())
    if overlap and not shared_frame:
        runner.chain.set_overlap(0)  # the sweep times every kernel with nothing beside it
    ktimes = kernel_sweep(runner, kernels) if not args.no_kernel_sweep else {}
    if overlap and not shared_frame:
        runner.chain.set_overlap(overlap)
        if ktimes:
            for i in range(4):
                runner.step()  # (the two streams back in their steady state)
    dominant = max(ktimes, key=ktimes.get) if ktimes else None
    if rank == 0 and dominant:
        runner.arm_kernel_timing(dominant, args.steps)  # HIP events around every launch of the dominant kernel inside the timed region
    lib_comm = getattr(runner, "mifx_comm", None) if shared_frame else None
    comm0 = lib_comm.stats() if lib_comm is not None else None  # (host counters: bytes handed to the transport so far)
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # one event per frame boundary: the median frame beside the mean
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        runner.step()
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = marks[0].elapsed_time(marks[-1])
    frame_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = 0.5 * (frame_ms[(args.steps - 1) // 2] + frame_ms[args.steps // 2])
    from diligentfx_amd.dist import max_over_ranks

    elapsed = max_over_ranks(elapsed, side_dev)  # the slowest rank defines the step time (covered by tests/test_dist_gloo.py)
    comm1 = lib_comm.stats() if lib_comm is not None else None

    total_px = float(W) * H * (1 if shared_frame else world) * args.steps
    value = total_px / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    # output rows of THIS rank (ghost rows of the sharded mode are overhead, not work): from the actual cuts of a cost-weighted split, not H / world
    rows_gpu = (runner.cuts[rank + 1] - runner.cuts[rank]) if shared_frame and runner.cuts else H
    chain_gbs = chain_bpp * W * rows_gpu / (dev_ms / args.steps * 1e-3) / 1e9

    chain_workload = (f"full chain PBR+SSR+SSAO+composite+TAA+Bloom+ToneMap, one {W}x{H} frame row-band sharded over {world} GPUs (BASELINE configs[4] layout)" if shared_frame else
                      f"full chain PBR+SSR+SSAO+composite+TAA+{'DOF+' if args.dof else ''}Bloom+ToneMap {W}x{H} per GPU (BASELINE configs[3]{' + depth of field' if args.dof else ''})")
    workload = {"chain": chain_workload, "ssao1080": f"PostFX prep + SSAO (A2..A8 with temporal history) on a {W}x{H} synthetic depth + normal G-buffer (BASELINE configs[1])",
                "pbr4k": f"PBR GGX + IBL shade on a {W}x{H} synthetic G-buffer (albedo / normal / metal-rough / depth; radiance + specular-IBL targets) (BASELINE configs[2])"}[args.config]
    metric = {"chain": METRIC_CHAIN, "ssao1080": "Mpixels/s SSAO @1080p (BASELINE configs[1]); %HBM roofline",
              "pbr4k": "Mpixels/s PBR GGX+IBL shade @4K (BASELINE configs[2]); %HBM roofline"}[args.config]
    result = {
        "metric": metric,
        "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_frames_run": warmup_frames,
        "ms_per_step": round(ms_per_step, 4), "ms_per_step_median": round(median_ms, 4), "higher_is_better": True, "scaling": "strong" if shared_frame else "weak", "vs_baseline": None,
        "dtype": "f32" if args.storage == "fp32" else "f32 arithmetic, storage in the reference's target formats (RGBA16F / R8 / R16F / RG16F / R11G11B10F)", "data": "synthetic",
        "config": {"workload": workload, "width": W, "height_per_gpu": rows_gpu,
                   "sharding": runner.sharding_note(), "storage": "fp32 planes" if args.storage == "fp32" else "libmifx_h4.so: RGBA16_FLOAT colour planes, R8_UNORM AO / roughness, R16_FLOAT variance / history length, RG16_FLOAT closest motion, R11G11B10_FLOAT Bloom levels; fp32 depth",
                   "taa": "bicubic", "ssao": "GTAO half-res + bilateral upsampling" if args.ssao_half else "GTAO full-res", "ssr": "half-res rays" if args.ssr_half else "full-res rays", "tonemap": "Uncharted2+sRGB",
                   "ibl": "static maps: the shade's apron copy is made once (mifx_postfx_set_static_ibl)", "fusion_mask": fusion_mask,
                   "fp_policy": "parity first: no FMA contraction in any source, separate multiplies and adds in the SSR march (libmifx.so as built by build.py; the contracting "
                                "build was 1.0 % faster and left 2.5e-4 .. 1.4e-3 of the shade / TAA / ray-march values beyond 1e-3: profiles/r04_ab_nofma_vs_fast.txt, r04_parity_outliers_default_build.txt)",
                   "stream_overlap": {0: "none (one stream)", 1: "prep + SSAO on a second stream beside shade + SSR", 2: "prep + SSAO on a second stream, across frames (mifx_chain_set_overlap 2)",
                                      3: "three lanes across frames: shade + prep + Hi-Z + SSAO | SSR + composite + TAA | Bloom + tone map (mifx_chain_set_overlap 3)",
                                      4: "three lanes, two frames in flight: shade + prep + Hi-Z + SSAO of frame N + 1 beside SSR + composite + TAA of frame N, Bloom + tone map of frame N - 1 "
                                         "(mifx_chain_set_overlap 4)" + (f"; lane edges {args.lane_edges}" if args.lane_edges else ""),
                                      5: "three lanes, two frames in flight, the bandwidth-bound tail on the Bloom lane: shade + prep + Hi-Z + SSAO | SSR R4 .. R6 | composite + TAA + Bloom + "
                                         "tone map (mifx_chain_set_overlap 5)"}[overlap]
                                     if not (shared_frame and getattr(runner, "mifx_comm", None) is not None) else
                                     {2: "sharded frame as two lanes across frames: phases 0 - 2 (shade .. TAA, Bloom's fine levels, the exchanges) | phase 3 (Bloom's coarse levels, final pass) "
                                         "beside the next frame's shade and SSAO (mifx_chain_set_overlap 2 under mifx_chain_execute_sharded)",
                                      3: "sharded frame as three lanes across frames: shade, SSR, composite, TAA, Bloom's fine levels, the exchanges | PostFX prep + SSAO | phase 3 (Bloom's coarse "
                                         "levels, final pass) beside the next frame (mifx_chain_set_overlap 3 under mifx_chain_execute_sharded)"}.get(
                                         int(os.environ.get("MIFX_SHARD_OVERLAP", "3")), "sharded frame, MIFX_SHARD_OVERLAP=" + os.environ.get("MIFX_SHARD_OVERLAP", "3")),
                   "chain_algorithmic_bytes_per_px": round(chain_bpp, 1), "chain_hbm_frac": round(chain_gbs / HBM_PEAK_GBS, 4)},
    }

    # ---------------------------------------------------------------- roofline of the dominant kernel (HIP events over the timed region)
    if rank == 0 and dominant:
        kt = runner.kernel_times_ms(args.steps)
        runner.arm_kernel_timing(None, 0)
        k_ms = sum(kt) / max(len(kt), 1)
        px = W * rows_gpu

        def roof(name, ms):
            algo = kernels[name] * px
            ach = algo / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            traffic, chain_traffic, src = pmc_traffic(W, H, name) if not shared_frame else (None, None, None)
            return {"kernel": name, "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": round(algo),
                    "kernel_ms": round(ms, 5)}, chain_traffic, src

        # `frac` follows the kernel's own duration (nothing beside it: what a rocprofv3 kernel trace of the one-stream frame reports, profiles/r05_kernel_stats_*.txt); the
        # HIP-event bracket inside the timed region of a multi-stream mode also holds the wait for wave slots other lanes' kernels occupy: reported beside it
        alone_ms = ktimes.get(dominant, 0.0) if overlap else k_ms
        dom, chain_traffic, src = roof(dominant, alone_ms if alone_ms > 0 else k_ms)
        in_region, _, _ = roof(dominant, k_ms)
        fracs = {n: kernels[n] * px / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS for n, ms in ktimes.items() if ms > 0}
        lowest_name = min(fracs, key=fracs.get)
        lowest, _, _ = roof(lowest_name, ktimes[lowest_name])
        lowest["measured"] = "untimed sweep, 3 launches"
        copy_gbs = measured_copy_peak(runner, dev, torch)
        result["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": dom["frac"], "traffic": dom["traffic"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                              "kernel_ms": dom["kernel_ms"], "launches_timed": len(kt),
                              "kernel_ms_in_timed_region": in_region["kernel_ms"], "achieved_including_queueing": in_region["achieved"], "frac_including_queueing": in_region["frac"],
                              "note": ("kernel_ms / achieved / frac: the kernel's own duration, HIP events around it with nothing beside it (this run's untimed one-stream sweep, 3 launches; agrees "
                                       "with the rocprofv3 kernel trace of the one-stream frame under profiles/); kernel_ms_in_timed_region / frac_including_queueing: HIP events around every "
                                       "launch inside the timed region, where the other lanes' kernels share the GPU with it -- that bracket opens when the lane reaches the launch, so it "
                                       "also holds the wait for the wave slots those kernels occupy") if overlap else None,
                              "traffic_source": (src + " (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE, separate passes; corrections in the file)") if dom["traffic"] else None,
                              "traffic_matches_build": (traffic_file()[2] or {}).get("matches") if not shared_frame else None,
                              "traffic_build_identity": traffic_file()[2] if not shared_frame else None,
                              "achievable_peak_measured": round(copy_gbs, 1),
                              "achievable_peak_how": "mifx_debug_stream_copy: 1 GiB device-to-device, one 16-byte texel per lane (read + write bytes, median of 10); the guide's figure is ~6.3 TB/s",
                              "selection": "the bracketed kernel with the longest average duration in this run's own untimed sweep",
                              "lowest": lowest,
                              "per_kernel_ms": {n: round(ms, 4) for n, ms in sorted(ktimes.items(), key=lambda kv: -kv[1])},
                              "per_kernel_frac": {n: round(f, 4) for n, f in sorted(fracs.items(), key=lambda kv: kv[1])},
                              "per_kernel_frac_measured_bytes": measured_byte_fractions(W, H, ktimes) if not shared_frame else None,
                              "whole_chain": {"achieved": round(chain_gbs, 1), "frac": round(chain_gbs / HBM_PEAK_GBS, 4), "traffic": chain_traffic if not stage else None,
                                              "algorithmic_bytes": round(chain_bpp * W * rows_gpu), "frac_of_achievable": round(chain_gbs / copy_gbs, 4)}}
        valu = valu_roof(W, H, ktimes)
        if valu:
            result["roofline"]["valu"] = valu
        if not shared_frame and not stage:
            sol = speed_of_light(W, H, ktimes, copy_gbs)
            if sol:
                result["roofline"]["speed_of_light"] = sol
        # per-stage sweep (separate frames, stage events of the chain; serial streams)
        if not args.no_pass_breakdown and not shared_frame and not stage:  # (the sharded mode steps all ranks together: no rank-0-only frames)
            passes = runner.time_passes(reps=10)
            result["roofline"]["per_pass_ms"] = {k: round(v["ms"], 4) for k, v in passes.items()}
            result["roofline"]["per_pass_frac"] = {k: round(v["algo_bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for k, v in passes.items() if v["algo_bytes"]}

    # ---------------------------------------------------------------- the run's own stream mode against the one-stream chain, bit for bit, at the run's own size
    if world == 1 and not stage and overlap and not args.no_overlap_check:
        try:
            n_cmp = 6
            n_bad = runner.verify_overlap_against_one_stream(n_cmp, fusion_mask=args.fusion_mask)
            result["overlap_verified"] = {"mode": overlap, "frames_compared": n_cmp, "frames_that_differed": n_bad,
                                          "how": f"after the timed region: {n_cmp} frames of this run's chain (mifx_chain_set_overlap {overlap}) queued back to back from a history reset, each into its "
                                                 f"own plane, against a second chain object on one stream (mode 0) on the same orbit positions at {W}x{H}: torch.equal per frame"}
        except Exception as e:  # noqa: BLE001
            result["overlap_verified"] = {"mode": overlap, "failed": repr(e)}

    # ---------------------------------------------------------------- BASELINE configs[1] and [2] inside the default line (measured after the timed region, ~1 s each)
    if rank == 0 and world == 1 and not stage and not shared_frame and not args.no_stage_lines and args.storage == "fp32":
        try:
            result["config"]["stage_lines"] = stage_lines(local_rank, tables, torch, with_cpu=not args.no_cpu_baseline, layers=args.layers_line)
        except Exception as e:  # extra information, never a reason to lose the bench line
            result["config"]["stage_lines"] = {"failed": repr(e)}

    # ---------------------------------------------------------------- CPU baseline: the oracle / reference on the host cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline_stage("ssao" if args.config == "ssao1080" else "pbr", (W, H), dev) if stage else cpu_baseline(budget_s=20.0, size=(W, H), device=dev)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
            result["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}

    if calibration is not None:
        result["config"]["band_calibration"] = {"rounds": calibration, "final_cuts": list(runner.cuts),
                                                "how": "each rank's band timed without the exchanges, all-gathered, tiling.refine_cuts; histories reset afterwards"}
    if shared_frame:
        # ---- what carried the rows: the transport, RCCL's own rank count, bytes per frame (host counters over the timed region) and the exchange groups' durations (HIP
        # events around every group of three extra frames after the timed region: the timed frames themselves carry no extra event records)
        comm = runner.comm_report()
        if lib_comm is not None:
            comm["bytes_sent_per_frame"] = (comm1["bytes_sent"] - comm0["bytes_sent"]) // args.steps
            comm["bytes_received_per_frame"] = (comm1["bytes_received"] - comm0["bytes_received"]) // args.steps
            comm["exchange_groups_per_frame"] = (comm1["groups"] - comm0["groups"]) / args.steps
            lib_comm.set_timing(True)
            for _ in range(3):
                runner.step()
            torch.cuda.synchronize()
            st = lib_comm.stats()
            lib_comm.set_timing(False)
            comm["exchange_ms"] = {"per_frame": round(st["exchange_ms_total"] / 3.0, 4), "longest_group": round(st["exchange_ms_max"], 4), "groups_timed": st["timed_groups"], "frames": 3,
                                   "how": "HIP events on the stream each group is issued on (SSAO halos | Bloom gather | TAA + SSR halos), start = the stream reaches the group, stop = its "
                                          "transfers are done on this rank (a late peer's delay included); the halo groups run on their own stream beside compute, so their sum is not frame time"}
        comm["rank"] = 0
        result["comm"] = comm
        # ---- the same frame, whole, on ONE GPU (rank 0's, the others wait): what the sharded frame time compares with -- the N = 1 line of this bench times a 3840x2160
        # frame, a quarter of the pixels, and is not that reference
        single_ms = None
        if rank == 0 and not args.no_single_gpu_reference:
            try:
                single_ms = runner.time_unsharded_same_frame(frames=max(6, min(args.steps, 20)), warm=6)
            except Exception as e:  # noqa: BLE001 -- extra information, never a reason to lose the line
                result["single_gpu_same_frame_failed"] = repr(e)
        barrier()
        if single_ms is not None:
            result["single_gpu_same_frame_ms"] = round(single_ms, 4)
            result["speedup_vs_single_gpu_same_frame"] = round(single_ms / ms_per_step, 3)
            result["single_gpu_same_frame_how"] = (f"the unsharded chain (mifx_chain_execute, three lanes with two frames in flight: mifx_chain_set_overlap 5) on the whole {W}x{H} frame of the same orbit on rank 0's GPU "
                                                   "after the timed region, the other ranks idle" + ("; with --single-gpu every rank shares that GPU, so the ratio is not a scaling figure" if args.single_gpu else ""))
        # the sharded frame against the unsharded chain, bit for bit: every frame of the run with --verify-shard, else three frames after the timed region
        if args.verify_shard and runner.mifx_comm is None:
            n_cmp, n_bad = args.warmup + args.steps, runner.mismatches
        else:
            n_cmp = 3
            n_bad = runner.verify_against_unsharded(n_cmp)
        bad = torch.tensor([n_bad], dtype=torch.int64, device=side_dev)
        dist.all_reduce(bad)
        result["shard_verified"] = {"frames_compared": n_cmp, "bands_that_differed": int(bad.item()),
                                    "how": "every rank also runs the unsharded chain from the same history reset and compares its band of the output bit for bit"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if watchdog is not None:
        watchdog.cancel()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
