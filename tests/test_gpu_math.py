"""The kernels' fp32 helper functions against IEEE / double precision ("fp policy"), through mifx_debug_eval_math.

fdiv   : rcp + one FMA residual step + v_div_fixup   -- must equal the IEEE quotient except ~1 in 4 million (by 1 ulp), incl. 0 / inf / NaN operands
fsqrt  : hardware estimate + next-up/down residual test -- must equal the IEEE square root exactly
sincos : Cody-Waite + Cephes polynomials for bounded angles -- <= 2 ulp, the accuracy class of libm's sinf / cosf
exp/pow: hardware exp2 / log2 -- smooth weights and the sRGB curve, ~1e-6 relative
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FDIV, FSQRT, SIN, COS, EXP, POW = range(6)


def eval_math(op, a, b=None):
    from diligentfx_amd import api, binding as B

    ctx = api.PostFXContext(0)
    ta = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(ctx.device)
    tb = torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(ctx.device) if b is not None else None
    out = torch.empty_like(ta)
    fp = ctypes.POINTER(ctypes.c_float)
    lib = B.load()
    lib.mifx_debug_eval_math.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    lib.mifx_debug_eval_math.restype = ctypes.c_int
    B.check(lib.mifx_debug_eval_math(ctx.handle, op, ta.data_ptr(), tb.data_ptr() if tb is not None else None, out.data_ptr(), ta.numel()))
    torch.cuda.synchronize()
    res = out.cpu().numpy()
    ctx.close()
    del fp
    return res


def ulp_diff(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-(2 ** 31)) - ia, ia)
    ib = np.where(ib < 0, np.int64(-(2 ** 31)) - ib, ib)
    return np.abs(ia - ib)


def test_fdiv_matches_ieee_division(mifx_lib):
    rng = np.random.default_rng(7)
    n = 1 << 23
    # magnitudes over 24 binades each, both signs: the operand ranges of the path (depths, sizes, weights, colour) and well beyond
    a = (np.exp2(rng.uniform(-12, 12, n)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    b = (np.exp2(rng.uniform(-12, 12, n)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    got, want = eval_math(FDIV, a, b), a / b
    d = ulp_diff(got, want)
    assert d.max() <= 1, f"fdiv off by {d.max()} ulp"
    frac = float((d != 0).mean())
    assert frac < 2e-6, f"fdiv differs from IEEE division for {frac:.2e} of the quotients (expected ~2.4e-7)"


def test_fdiv_special_operands(mifx_lib):
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    a = np.array([1, -1, 0, 0, inf, -inf, inf, 3, nan, 1, 0.0, 5, -0.0], np.float32)
    b = np.array([0, 0, 0, 5, 2, 2, inf, inf, 1, nan, -0.0, -0.0, 7], np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        want = a / b
    got = eval_math(FDIV, a, b)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (got, want)
    m = ~np.isnan(want)
    assert np.array_equal(got[m], want[m]) and np.array_equal(np.signbit(got[m]), np.signbit(want[m])), (got, want)


def test_fsqrt_is_correctly_rounded(mifx_lib):
    rng = np.random.default_rng(8)
    n = 1 << 23
    x = np.exp2(rng.uniform(-40, 40, n)).astype(np.float32)
    x[:8] = [0.0, 1.0, 4.0, 2.0, np.inf, 1e-30, 3.0, 0.25]
    got, want = eval_math(FSQRT, x), np.sqrt(x)
    assert np.array_equal(got, want), f"fsqrt differs from IEEE sqrt in {(got != want).sum()} of {n} values"
    assert np.isnan(eval_math(FSQRT, np.array([-1.0, np.nan], np.float32))).all()


@pytest.mark.parametrize("op,fn", [(SIN, np.sin), (COS, np.cos)])
def test_bounded_sincos(mifx_lib, op, fn):
    x = np.linspace(-13.0, 13.0, 1 << 22).astype(np.float32)
    got, want64 = eval_math(op, x), fn(x.astype(np.float64))
    want = want64.astype(np.float32)
    assert np.abs(got - want64).max() < 1.5e-7
    big = np.abs(want64) > 1e-3  # ulp distance is only meaningful away from the zeros of the function
    assert ulp_diff(got[big], want[big]).max() <= 2


def test_hardware_exp_pow(mifx_lib):
    rng = np.random.default_rng(9)
    x = rng.uniform(-20.0, 5.0, 1 << 20).astype(np.float32)
    got = eval_math(EXP, x)
    assert (np.abs(got - np.exp(x.astype(np.float64))) / np.exp(x.astype(np.float64))).max() < 5e-6
    a = np.exp2(rng.uniform(-10, 4, 1 << 20)).astype(np.float32)
    e = rng.uniform(0.2, 3.0, 1 << 20).astype(np.float32)
    want = np.power(a.astype(np.float64), e.astype(np.float64))
    assert (np.abs(eval_math(POW, a, e) - want) / want).max() < 2e-5
