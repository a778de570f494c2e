#!/usr/bin/env python3
"""Round 5 probe: does a CU-masked stream (hipExtStreamCreateWithCUMask) change what two kinds of work cost side by side?
tools/exp_pairs.py (round 4) measured plain streams: gather work (SSAO) beside a pure HBM copy hides 22 % of the copy -- the hardware shares the CUs as it likes.
An HBM-bound kernel needs only a fraction of the CUs to reach its rate; the L1- / ALU-bound kernels scale with the CUs they get.  So: the copy on n CUs, the gather
work on all of them or on the others.
   S = PostFX prep + SSAO A2..A8        P = the PBR shade        C = a device-to-device copy (~1 GB of traffic)
Mask layouts: "first" = the low n bits; "spread" = 8 consecutive bits of every 32 (balanced whether the driver deals mask bits round-robin to the XCDs or in blocks)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import binding as B, tiling  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def mask_words(kind, n):
    bits = [0] * 256
    if kind == "first":
        for i in range(n):
            bits[i] = 1
    elif kind == "last":
        for i in range(256 - n, 256):
            bits[i] = 1
    elif kind == "spread":      # n / 64 of the four 8-bit groups of every 32-bit word
        rows = n // 64
        for i in range(256):
            if (i // 8) % 4 < rows:
                bits[i] = 1
    elif kind == "spread_hi":   # the complement layout: the LAST rows of every word
        rows = n // 64
        for i in range(256):
            if (i // 8) % 4 >= 4 - rows:
                bits[i] = 1
    words = (ctypes.c_uint32 * 8)()
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= 1 << (i % 32)
    return words


def masked_stream(kind, n):
    if kind == "full":
        return torch.cuda.Stream()
    s = ctypes.c_void_p()
    w = mask_words(kind, n)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(8), w)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask({kind}, {n}) -> {rc}")
    back = (ctypes.c_uint32 * 8)()
    rc = hip.hipExtStreamGetCUMask(s, ctypes.c_uint32(8), back)
    print(f"  stream {kind}{n}: asked {' '.join(f'{x:08x}' for x in w)}  got(rc {rc}) {' '.join(f'{x:08x}' for x in back)}", flush=True)
    return torch.cuda.ExternalStream(s.value)


tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
W, H, K = 3840, 2160, 40
ssao = tiling.StageRunner("ssao", 0, tables["sobol_256d"], tables["scrambling_tile"], W, H)
pbr = tiling.StageRunner("pbr", 0, tables["sobol_256d"], tables["scrambling_tile"], W, H)
for r in (ssao, pbr):
    r.build_inputs(n_frames=6)
dev = ssao.dev
n = W * H * 4 * 4
src, dst = torch.ones(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
copy_runner = tiling.StageRunner("pbr", 0, tables["sobol_256d"], tables["scrambling_tile"], 64, 64)  # (kept alive: the context belongs to its chain)
copy_ctx = copy_runner.chain.postfx
lib = copy_ctx.lib


def step(k, stream, i):
    with torch.cuda.stream(stream):
        if k == "S":
            ssao.ctx.sync_stream()
            ssao.step(i)
        elif k == "P":
            pbr.ctx.sync_stream()
            pbr.step(i)
        else:
            copy_ctx.sync_stream()
            B.check(lib.mifx_debug_stream_copy(copy_ctx.handle, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), ctypes.c_uint64(4 * n)))


def run(active, steps=K):
    for i in range(8):
        for k, s in active:
            step(k, s, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        for k, s in active:
            step(k, s, 8 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


streams = {}


def st(kind, n=256):
    key = (kind, n)
    if key not in streams:
        streams[key] = masked_stream(kind, n)
    return streams[key]


print("== one kind of work alone, by the CUs its stream may use (ms per step; copy also as TB/s of read + write)")
alone = {}
for kind, cus in (("full", 256), ("first", 64), ("spread", 64), ("first", 128), ("spread", 128), ("spread", 192), ("spread_hi", 192), ("spread_hi", 128)):
    for k in "CSP":
        t = run([(k, st(kind, cus))])
        alone[(k, kind, cus)] = t
        extra = f"  {2 * 4 * n / t / 1e9:.2f} TB/s" if k == "C" else ""
        print(f"  {k} on {kind}{cus}: {t:.3f}{extra}", flush=True)

print("== pairs (ms per pair of steps; hidden = share of the shorter one's ALONE-ON-ALL-CUs time that disappeared)")
full = {k: alone[(k, "full", 256)] for k in "CSP"}
for (ka, kinda, na), (kb, kindb, nb) in (
    (("S", "full", 256), ("C", "full", 256)),
    (("S", "full", 256), ("C", "spread", 64)),
    (("S", "full", 256), ("C", "spread", 128)),
    (("S", "spread_hi", 192), ("C", "spread", 64)),
    (("S", "spread_hi", 128), ("C", "spread", 128)),
    (("P", "full", 256), ("C", "full", 256)),
    (("P", "full", 256), ("C", "spread", 64)),
    (("P", "spread_hi", 192), ("C", "spread", 64)),
    (("P", "spread_hi", 128), ("C", "spread", 128)),
    (("S", "full", 256), ("P", "full", 256)),
    (("S", "spread_hi", 128), ("P", "spread", 128)),
    (("S", "spread_hi", 192), ("P", "spread", 64)),
    (("S", "spread", 64), ("P", "spread_hi", 192)),
):
    t = run([(ka, st(kinda, na)), (kb, st(kindb, nb))])
    hidden = (full[ka] + full[kb] - t) / min(full[ka], full[kb])
    print(f"  {ka}@{kinda}{na} beside {kb}@{kindb}{nb}: {t:.3f}; sum alone (all CUs) {full[ka] + full[kb]:.3f}; hidden {hidden * 100:.0f} %", flush=True)
