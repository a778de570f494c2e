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
    assert bench.DOF_BPP > 100 and bench.ROOFLINE_KERNEL == "ssr_intersection_kernel"
