"""Host-side pieces of bench.py that do not need a GPU: the CPU-baseline leg (the checker timed on the host cores) on a tiny sample."""
import importlib.util
import os

from util import ROOT


def load_bench():
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cpu_baseline_leg_runs():
    bench = load_bench()
    from diligentfx_amd import synth

    calls, orig = [], synth.make_frame

    def counting(*a, **k):
        calls.append(a[1])
        return orig(*a, **k)

    synth.make_frame = counting
    try:
        r = bench.cpu_baseline(budget_s=0.05, size=(96, 64))
    finally:
        synth.make_frame = orig
    assert len(calls) == len(set(calls)), "a frame was rendered twice: the input cache of the timed region missed"
    assert r["unit"] == "Mpixels/s" and r["value"] > 0 and r["kind"] in ("reference", "port") and r["cores"] >= 1 and "96x64" in r["sample"]


def test_algorithmic_bytes_tables_agree():
    from diligentfx_amd import tiling

    bench = load_bench()
    assert bench.ALGO_BPP == tiling.ALGO_BPP and abs(bench.CHAIN_BPP - sum(bench.ALGO_BPP.values())) < 1e-9
    assert bench.DOF_BPP > 100
    # the per-kernel table (the roofline kernel is picked from it by measurement) sums to the stage figures of SURVEY Appendix C
    k = bench.KERNEL_BPP
    assert abs(k["ssr_mask_roughness_kernel"] + k["ssr_intersection_kernel"] + k["ssr_spatial_kernel"] + k["ssr_temporal_kernel"] + k["ssr_bilateral_kernel"] + 5.33 - 327.7) < 0.1
    assert abs(k["ssao_compute_ao_kernel"] + k["ssao_temporal_kernel"] + k["ssao_resample_kernel"] + k["ssao_spatial_kernel"] + 5.33 + 10.67 - 148.0) < 0.1
    assert k["pbr_shade_kernel"] == 84.0 and k["composite_kernel"] == 116.0 and k["taa_kernel"] == 64.0 and k["tonemap_kernel"] == 32.0 and k["postfx_prep_kernel"] == 28.0


def test_usable_cores_and_orbit_walk():
    bench = load_bench()
    n, desc = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and "physical" in desc
    # the forwards-and-back walk over the resident orbit: consecutive positions differ by one step, every position is visited
    from diligentfx_amd import tiling

    class Fake:
        frames = [None] * 5
    walk = [tiling.TiledChain.orbit_position(Fake, t) for t in range(20)]
    assert [k for k, _ in walk[:9]] == [0, 1, 2, 3, 4, 3, 2, 1, 0] and all(abs(k - kp) <= 1 for k, kp in walk) and all(kp == walk[i - 1][0] for i, (k, kp) in enumerate(walk) if i)


def test_band_cuts_cover_the_frame_and_balance_the_modelled_cost():
    """tiling.cost_weighted_cuts: bands partition the rows, respect min_rows, and (three-class model, ghost rows included) no band's modelled cost exceeds the
    others' by more than one row's worth -- on a frame of sky above, plain geometry in the middle and reflection samples below."""
    import torch

    from diligentfx_amd import tiling

    h, w, world = 1200, 64, 8
    depth = torch.ones(h, w)
    rough = torch.ones(h, w)
    depth[400:] = 0.5            # geometry from row 400 down
    rough[800:] = 0.1            # reflection samples from row 800 down
    for kw in ({}, {"sky_cost": tiling.SKY_COST, "roughness": rough, "reflective_cost": tiling.REFLECTIVE_COST}):
        cuts = tiling.cost_weighted_cuts(depth, world, 32, **kw)
        assert cuts[0] == 0 and cuts[-1] == h and len(cuts) == world + 1
        assert all(b - a >= 32 for a, b in zip(cuts, cuts[1:]))
        assert cuts == tiling.cost_weighted_cuts(depth, world, 32, **kw)  # deterministic: every rank computes the same cuts
    # the modelled cost of the three-class bands (own + 60 ghost rows each side)
    wrow = torch.where(depth[:, 0] >= 1.0, torch.tensor(tiling.SKY_COST), torch.where(rough[:, 0] <= 0.2, torch.tensor(tiling.REFLECTIVE_COST), torch.tensor(1.0))).double()
    cum = torch.cat([torch.zeros(1, dtype=torch.float64), torch.cumsum(wrow, 0)])
    cost = [float(cum[min(b + 60, h)] - cum[max(a - 60, 0)]) for a, b in zip(cuts, cuts[1:])]
    assert max(cost) - min(cost[:-1]) <= 2.0 * tiling.REFLECTIVE_COST + 1e-9, cost  # (the last band takes what is left: it may be cheaper)
    assert cuts[1] > h // world  # the sky band is taller than an equal split


def test_stage_bytes_move_with_the_fused_passes():
    """bench.stage_bytes: a fused pass is counted in the stage it runs in; the chain's total does not change and no stage is left with bytes it does not move."""
    bench = load_bench()
    total = sum(bench.ALGO_BPP.values())
    for mask in range(16):
        b = bench.stage_bytes(bench.ALGO_BPP, bench.KERNEL_BPP, mask)
        assert abs(sum(b.values()) - total) < 1e-9 and all(v >= 0 for v in b.values())
        assert (b["tonemap"] == 0.0) == bool(mask & 1)
        assert abs(b["pbr_shade"] - (84.0 + (25.0 if mask & 2 else 0.0))) < 1e-9 and abs(b["composite"] - (116.0 + (61.0 if mask & 4 else 0.0))) < 1e-9


def test_stage_cpu_baselines_run():
    bench = load_bench()
    for mode in ("ssao", "pbr"):
        r = bench.cpu_baseline_stage(mode, (96, 64), None, budget_s=0.02)
        assert r["value"] > 0 and r["unit"] == "Mpixels/s" and "96x64" in r["sample"] and r["kind"] in ("reference", "port")


def test_valu_roof_reads_the_committed_measurements(tmp_path):
    """bench.valu_roof: SQ_INSTS_VALU per dispatch x the measured issue time of a wave64 v_fma_f32 / (1024 SIMDs x kernel time)."""
    bench = load_bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r09_pmc_sq_counters_x.txt").write_text(
        "kernel                                                disp   dur_us           SQ_WAVES      SQ_INSTS_VALU   (per dispatch)\n"
        "mifx::ssr_intersection_kernel<false, false>              7    355.7           129600.0        199049954.3\n"
        "mifx::taa_kernel<false, true, false>                     7    175.4           129600.0        106594255.7\n")
    (prof / "r09_valu_issue_rate_x.txt").write_text("header\nv_fma_f32                                  W=8     51.2 /    51.3 ms    1.250 /  1.270 ns per wave-instruction per SIMD (spread  1.6 %)\n")
    bench.ROOT = str(tmp_path)
    r = bench.valu_roof(3840, 2160, {"ssr_intersection_kernel": 0.3, "taa_kernel": 0.16, "unknown_kernel": 1.0})
    assert abs(r["issue_ns"] - 1.26) < 1e-9 and set(r["per_kernel"]) == {"ssr_intersection_kernel", "taa_kernel"}
    k = r["per_kernel"]["ssr_intersection_kernel"]
    assert abs(k["insts_per_px"] - 199049954.3 * 64 / (3840 * 2160)) < 0.1 and abs(k["frac"] - 199049954.3 * 1.26e-9 / 1024 / 0.3e-3) < 1e-3
    assert bench.valu_roof(1920, 1080, {"taa_kernel": 0.1}) is None  # the counters were taken at another resolution


def _run_bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=300)
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    return r, lines


def test_plain_python_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) re-executes itself as two ranks under torch.distributed.run: the ranks join a process group
    and rank 0 prints ONE line with n_gpus 2 (round 5 silently timed one GPU).  --dry-run-ranks: the plumbing without the frames, so that it runs without a GPU."""
    r, lines = _run_bench(["--gpus", "2", "--dry-run-ranks"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["launcher"] == "self" and d["backend"] == "gloo" and sorted(x[0] for x in d["ranks"]) == [0, 1]
    assert len({x[2] for x in d["ranks"]}) == 2  # two processes


def test_bench_refuses_a_world_that_is_not_gpus():
    """Inside a launcher whose world size is not --gpus, and on a node with fewer GPUs than ranks: an {"error": ...} line and a non-zero exit code -- never a one-GPU
    measurement labelled otherwise."""
    r, lines = _run_bench(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and len(lines) == 1 and "error" in lines[0] and lines[0]["value"] is None and lines[0]["n_gpus"] == 8, (r.returncode, r.stdout)
    import torch

    if torch.cuda.device_count() < 2:  # (this container: no GPU at all)
        r, lines = _run_bench(["--gpus", "2"])
        assert r.returncode != 0 and len(lines) == 1 and "error" in lines[0] and lines[0]["gpus_visible"] == torch.cuda.device_count(), (r.returncode, r.stdout)


def test_a_launcher_that_dies_still_leaves_one_line(tmp_path, monkeypatch):
    """launch_ranks(): ranks that end without a result line -> the launcher prints the error line itself and returns their exit code."""
    import io
    import json

    bench = load_bench()
    args = bench.parse_args(["--gpus", "2", "--dry-run-ranks", "--backend", "no_such_backend"])
    out = io.StringIO()
    monkeypatch.setattr(bench.sys, "stdout", out)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.launch_ranks(args, ["--gpus", "2", "--dry-run-ranks", "--backend", "no_such_backend"])
    lines = [json.loads(x) for x in out.getvalue().splitlines() if x.startswith("{")]
    assert rc != 0 and len(lines) == 1 and "error" in lines[0] and "exit code" in lines[0]["error"]


def test_a_rank_that_never_finishes_does_not_hang_the_node():
    """--watchdog-s: rank 1 of two hangs (after the line of the dry run has been printed by rank 0) -> it gives up with exit code 4 after the time limit and the launcher
    returns a non-zero code instead of waiting for ever."""
    import time

    t0 = time.time()
    r, lines = _run_bench(["--gpus", "2", "--dry-run-ranks", "--dry-run-hang-rank", "1", "--watchdog-s", "8"])
    assert r.returncode != 0 and time.time() - t0 < 120, (r.returncode, time.time() - t0)
    assert "watchdog: rank 1 of 2" in r.stderr, r.stderr[-1500:]


def test_bench_arguments_of_round_6():
    """--cuts replays recorded band cuts (and switches the calibration off in main), --backend defaults to the gloo side channel beside the library's own communicator,
    --exact-warmup is what the profiling recipes pass, the watchdog is on by default for N > 1."""
    bench = load_bench()
    a = bench.parse_args(["--gpus", "8", "--cuts", "0,1358,1919,2380,2750,3140,3536,3917,4320"])
    assert a.cuts == [0, 1358, 1919, 2380, 2750, 3140, 3536, 3917, 4320] and a.backend is None and a.comm == "rccl" and a.watchdog_s == 900.0 and not a.exact_warmup
    assert bench.parse_args(["--exact-warmup", "--watchdog-s", "0"]).exact_warmup
    assert bench.rank_environment() is None or "WORLD_SIZE" in os.environ
