"""The product's HOST code on the CPU: diligentfx_amd/csrc/api_*.cpp + mifx_core.cpp exactly as they ship, linked with a stand-in for the HIP runtime and with launchers that
run the reference's own shaders (oracle/_ref) for each pass (tests/cpu_product/).  The effect objects are then driven through the C ABI by the very functions of the device test
(tests/test_gpu_host_sequence.py) and compared with oracle/cpu_chain.py -- whose sequencing tests/test_host_sequence_vs_ref.py holds to the executed reference host classes --
for EQUALITY: the arithmetic on both sides is the same shader code, so any difference is a difference of sequencing (ping-pong slots, reset rules, alpha, clears, what a resize
or a flag change re-creates).  SURVEY 8a rows C0 / A0 / R0 / T0 / B0, on the product's side.  Test infrastructure: nothing under diligentfx_amd/ builds or loads any of it."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def cpu_lib():
    import importlib.util

    import pyref

    if pyref.ref_lib() is None:
        pytest.skip("needs oracle/_ref (the reference's shaders stand in for the kernels)")
    spec = importlib.util.spec_from_file_location("_cpu_product_build", os.path.join(HERE, "cpu_product", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    lib = m.build()
    if lib is None:
        pytest.skip("hipcc not available")
    return lib


def run(cpu_lib, *args, timeout=900, **extra_env):
    env = dict(os.environ, MIFX_LIB_PATH=cpu_lib, **extra_env)
    env.pop("MIFX_STORAGE", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "cpu_product", "run.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and "cpu product: done" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


def test_host_objects_on_the_cpu_equal_the_reference_sequencing(cpu_lib):
    out = run(cpu_lib, "scenarios")
    assert out.count("cpu product: scenario OK") >= 7, out


def test_depth_of_field_on_the_cpu_equals_the_reference_sequencing(cpu_lib):
    """TAA -> depth of field -> Bloom over ten steps (temporal CoC ping-pong, flag changes that re-create the targets, kernel changes, a resize)."""
    assert "cpu product: scenario OK: depth of field" in run(cpu_lib, "dof")


def test_random_sequences_on_the_cpu_product(cpu_lib):
    """Random sequences of 30 steps (resizes, frame-index moves, feature flags of every effect, AO algorithm, bokeh kernel, reversed depth, Bloom radius, resets, effects left
    out) -- the sequences of tests/test_gpu_host_sequence.py::test_random_sequences_through_the_c_abi, here compared for equality.  More seeds:
    MIFX_LIB_PATH=tests/cpu_product/_build/libmifx_cpu.so python tests/cpu_product/run.py random FIRST LAST."""
    out = run(cpu_lib, "random", "11", "13", timeout=1500)
    assert out.count("cpu product: random sequence OK") == 2, out


def test_the_chain_object_on_the_cpu_equals_the_cpu_chain(cpu_lib):
    """mifx_chain_execute -- the entry bench.py times -- with its default fusions (the shade writing SSR's mask planes, R7 inside the composite, the tone map inside Bloom's last
    pass): six frames and the replay after reset_history, plain and reversed depth, equal to the checker's chain bit for bit (the device test allows 5e-3 of the values to differ:
    flipped rays; here both sides run the same shaders, so nothing may)."""
    out = run(cpu_lib, "chain")
    assert out.count("cpu product: scenario OK: chain") == 2, out


def test_the_pipelined_chain_on_the_cpu_equals_the_cpu_chain(cpu_lib):
    """mifx_chain_set_overlap 4 (two frames in flight: the planes between lane S and lane X alternate between two sets, traded with the effect objects' own at the start of
    every frame) over the same frames, with lane edges set: which set a kernel is handed is host logic, so it shows here; the stream ordering does not (streams do nothing
    in this build: tests/test_gpu_chain.py holds that on the device)."""
    out = run(cpu_lib, "chain", MIFX_CHAIN_OVERLAP="4", MIFX_LANE_EDGES="ssao_compute_ao_kernel<ssr_intersection_kernel@1,taa_kernel<pbr_shade_ssr_mask_kernel@0")
    assert out.count("cpu product: scenario OK: chain") == 2, out
    out = run(cpu_lib, "chain", MIFX_CHAIN_OVERLAP="5")  # (round 6: mode 4 with the composite, TAA and depth of field on the Bloom lane)
    assert out.count("cpu product: scenario OK: chain") == 2, out


def test_argument_checks_of_the_round_5_entries(cpu_lib):
    """mifx_chain_set_lane_edges (well-formed and malformed lists), mifx_chain_set_overlap (0 .. 4), mifx_chain_set_fusion_mask (bit 5, refused beyond): on the CPU build's chain object."""
    assert "cpu product: scenario OK: arguments of the round-5 entries" in run(cpu_lib, "arguments")


def test_execute_band_on_the_cpu_equals_the_phases(cpu_lib):
    """mifx_chain_execute_band (api_comm.cpp execute_sharded_impl without a communicator: one stream, two lanes, three lanes requested) against mifx_chain_execute_phase 0 .. 4
    on a second chain object with the same band: band rows and histories equal, nothing written outside the band."""
    assert run(cpu_lib, "band", timeout=900).count("cpu product: execute_band OK") == 3


def test_the_order_of_the_lanes_has_no_unordered_pair(cpu_lib):
    """tests/cpu_product/order.py: the stand-in runtime keeps a vector clock per stream and event, the launch handlers report the planes they read and write, and every pair of
    conflicting accesses of two streams must have a happens-before edge.  mifx_chain_execute in every stream mode (0 - 4) with the default fusions, all and none, with depth of
    field; mifx_chain_execute_band under the sharded frame's two and three lanes; seven frames queued without a host synchronisation.  Control: every hipStreamWaitEvent of a
    steady-state frame dropped in turn -- each is either noticed or (one, without depth of field) guards a plane that configuration does not write."""
    out = run(cpu_lib, "order", timeout=1500)
    assert out.count("cpu product: order OK") == 24 and out.count("cpu product: order control") == 9, out  # (round 6: + mode 5 -- the composite / TAA on the Bloom lane)
    assert "overlap 3, depth of field: of the 5 waits of a steady-state frame, dropping 5 leaves an unordered pair" in out, out
    assert "overlap 3, band: of the 7 waits of a steady-state frame, dropping 7 leaves an unordered pair" in out, out  # (the SSAO lane and the depth-hierarchy lane)


def test_random_chain_sequences_keep_the_lanes_ordered(cpu_lib):
    """The random sequences of test_random_sequences_through_the_chain_object_on_the_cpu (sizes, frame indices, history resets, flag sets, the fusion mask and the stream mode
    changing from frame to frame) under order.py: the fills and re-allocations the library queues on the context's stream between frames against the lanes around them."""
    out = run(cpu_lib, "order_random", "0", "4", timeout=1500)
    assert out.count("cpu product: order OK: chain sequence") == 4, out


def test_random_sequences_through_the_chain_object_on_the_cpu(cpu_lib):
    """mifx_chain_execute over random sequences in which, beside sizes, frame indices, resets, TAA flag sets and the AO algorithm, the FUSION MASK and the stream-overlap mode
    change from frame to frame (tests/cpu_product/run.py chain_random): every frame equals the CPU chain."""
    out = run(cpu_lib, "chain_random", "0", "3", timeout=1500)
    assert out.count("cpu product: chain sequence OK") == 3, out


def test_the_chain_with_material_layers_on_the_cpu(cpu_lib):
    """mifx_chain_set_material_layers: four frames with all five layers and two shadow-mapped lights equal the CPU chain whose shade is the reference's permutation; switched off
    again, the chain equals one that never had layers (tests/test_gpu_pbr_layers.py::test_chain_with_material_layers, for equality)."""
    assert "cpu product: scenario OK: chain with material layers" in run(cpu_lib, "layers")


def test_row_bands_on_the_cpu_product(cpu_lib):
    """Row-band sharding of the chain (mifx_chain_execute_phase) on the CPU build: 2, 3 and 4 chain objects on equal and uneven bands (down to 8 rows: thinner than the history halos), odd pyramid
    sizes, half-resolution SSAO / SSR, depth of field, the exchanges done by copying rows between their planes -- the bands' rows equal the unsharded chain object's, rows outside a band are never written, the history
    planes are equal on band + halo.  The launch handlers write only the row window each launcher was given, so a pass reading rows nobody computed would show; the run also drops
    each of the two exchanges once and must see the bands differ."""
    out = run(cpu_lib, "sharded", "0", "9", timeout=2400)  # (case 8: the store windows of SSAO's depth pyramid levels bind)
    assert out.count("cpu product: sharded case OK") == 9 and out.count("exchange the bands differ, as they must") == 2, out


def test_execute_sharded_with_the_in_library_group_on_the_cpu(cpu_lib):
    """mifx_chain_execute_sharded over the communicator of one process (csrc/api_comm.cpp: the code that decides which rows travel to whom for the RCCL transport as well, with a
    copy in place of ncclSend / ncclRecv), one thread per rank: 2, 3 and 4 ranks, half resolution, 8-row bands whose ghost rows come from two ranks away, depth of field -- bands and
    history planes equal the unsharded chain object's (tests/test_comm.py::test_sharded_execute_in_process_group on the CPU build)."""
    out = run(cpu_lib, "local_group", "0", "7", timeout=2400)  # (case 6: mifx_chain_set_overlap 3, the SSAO lane's event requests)
    assert out.count("cpu product: in-library group OK") == 7, out


def test_blooms_level_0_halo_exchange_runs_and_shortens_the_history_halos(cpu_lib):
    """Round 6 (csrc/api_comm.cpp, mifx_bloom::halo_level0): in a sharded frame a rank prefilters the rows of Bloom's level 0 it owns and receives the rows beside its band's edges.
    Three ranks on uneven bands, the switch off and on (MIFX_SHARD_BLOOM_HALO, read per frame): one more exchange group per frame, fewer bytes per frame in total, shorter history
    halos reported by mifx_chain_get_shard_info -- and both frames equal the unsharded chain object's bit for bit."""
    assert "cpu product: Bloom level-0 halo OK" in run(cpu_lib, "bloom_halo", timeout=900)
