"""The host helpers of the boundary against the reference's own code (oracle/_ref: HaltonSequence, the arithmetic of GetJitterOffset, GetJitteredProjMatrix and
ReverseExpToneMap compiled from TemporalAntiAliasing.cpp:43-78, TemporalAntiAliasing.hpp:138-155 and Components/src/ToneMapping.cpp:43-83 where they lie,
oracle/ref/ref_t0_host_helpers.cpp).  Host arithmetic on both sides: bit-exact.  No GPU."""
import ctypes

import numpy as np
import pytest


def ref():
    import pyref

    r = pyref.ref_lib()
    if r is None or not r.has("ref_taa_jitter_offset"):
        pytest.skip("oracle/_ref with the host helpers is not available")
    return r


def test_jitter_offset_equals_reference(mifx_lib):
    r = ref()
    out = (ctypes.c_float * 2)()
    for w, h in ((1920, 1080), (3840, 2160), (7680, 4320), (333, 17), (1, 1)):
        for frame in list(range(0, 40)) + [255, 4096 + 7, 2 ** 32 - 1]:
            assert mifx_lib.mifx_taa_get_jitter_offset(ctypes.c_uint32(frame), ctypes.c_uint32(w), ctypes.c_uint32(h), out) == 0
            want = np.zeros((1, 1, 2), np.float32)
            r.call("ref_taa_jitter_offset", [], [want], ival=[frame if frame < 2 ** 31 else frame - 2 ** 32, w, h])
            assert np.array_equal(np.array([out[0], out[1]], np.float32), want[0, 0]), (frame, w, h, out[0], out[1], want)
    # the Halton sequence itself, base 2 / 3 / 5, beyond the 16-sample cycle
    from diligentfx_amd import synth

    for base in (2, 3, 5):
        for idx in range(0, 200):
            want = np.zeros((1, 1), np.float32)
            r.call("ref_halton_sequence", [], [want], ival=[base, idx])
            assert np.float32(synth.halton(base, idx)) == want[0, 0]


def test_jittered_proj_matrix_equals_reference(mifx_lib):
    r = ref()
    rng = np.random.default_rng(5)
    for case in range(20):
        proj = rng.standard_normal((4, 4)).astype(np.float32)
        if case % 2 == 0:
            proj[3, 3] = 0.0  # perspective: m33 == 0
        jitter = rng.standard_normal(2).astype(np.float32) * np.float32(1e-3)
        got = (ctypes.c_float * 16)()
        assert mifx_lib.mifx_taa_get_jittered_proj_matrix((ctypes.c_float * 16)(*proj.ravel()), (ctypes.c_float * 2)(*jitter), got) == 0
        want = np.zeros((4, 4), np.float32)
        r.call("ref_taa_jittered_proj_matrix", [proj], [want], fval=[float(jitter[0]), float(jitter[1])])
        assert np.array_equal(np.array(list(got), np.float32).reshape(4, 4), want), case


def test_reverse_exp_tone_map_equals_reference(mifx_lib):
    r = ref()
    rng = np.random.default_rng(6)
    ldr = np.concatenate([rng.random((200, 3)), np.zeros((1, 3)), np.full((1, 3), 0.5), rng.random((20, 3)) * 3.0, np.array([[0.995, 0.995, 0.995]])]).astype(np.float32)
    for mg, avg in ((0.18, 0.3), (0.5, 0.05), (0.18, 1.0)):
        want = np.zeros((1, ldr.shape[0], 3), np.float32)
        r.call("ref_reverse_exp_tone_map", [ldr[None]], [want], fval=[mg, avg])
        got = np.zeros_like(ldr)
        out = (ctypes.c_float * 3)()
        for i, c in enumerate(ldr):
            assert mifx_lib.mifx_reverse_exp_tone_map((ctypes.c_float * 3)(*c), ctypes.c_float(mg), ctypes.c_float(avg), out) == 0
            got[i] = out[:]
        assert np.array_equal(got, want[0]), (mg, avg, np.abs(got - want[0]).max())
