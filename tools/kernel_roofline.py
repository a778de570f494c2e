#!/usr/bin/env python3
"""Per-kernel HBM roofline table from the committed measurements: average duration (rocprofv3 --kernel-trace --stats, profiles/rNN_kernel_stats_*.txt) and HBM bytes per
frame (separate --pmc passes, profiles/rNN_pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE, calibrated on the tone-map kernel).

    python tools/kernel_roofline.py profiles/r01_kernel_stats_v7.txt profiles/r01_pmc_traffic.json > profiles/r01_roofline_per_kernel_v7.txt"""
import json
import sys

PEAK = 8000.0  # GB/s, MI355X HBM3E


def main():
    stats, traffic = sys.argv[1], json.load(open(sys.argv[2]))
    dur = {}
    for line in open(stats).read().splitlines()[2:]:
        f = line.split()
        if len(f) >= 4 and f[-3].isdigit():
            dur[" ".join(f[:-3]).replace("mifx::", "")] = (int(f[-3]), float(f[-2]), float(f[-1]))
    counts = [c for c, _, _ in dur.values()]
    frames = max(set(counts), key=counts.count)  # most kernels are launched once per frame
    print(f"per-kernel HBM roofline, {traffic['resolution'][0]}x{traffic['resolution'][1]}, build {traffic.get('build', '?')}; peak {PEAK:.0f} GB/s")
    print(f"{'kernel':46s} {'launches/frame':>14s} {'us/frame':>9s} {'HBM MB/frame':>13s} {'GB/s':>7s} {'% of peak':>9s}")
    rows, tot_us, tot_b = [], 0.0, 0.0
    for name, k in traffic["kernels"].items():
        if name not in dur:
            continue
        calls, total_us, _ = dur[name]
        us = total_us / frames
        b = k["read_bytes"] + k["write_bytes"]
        rows.append((us, name, calls / frames, b))
        tot_us += us
        tot_b += b
    for us, name, n, b in sorted(rows, reverse=True):
        gbs = b / (us * 1e-6) / 1e9
        print(f"{name:46s} {n:14.0f} {us:9.1f} {b / 1e6:13.1f} {gbs:7.0f} {100 * gbs / PEAK:8.1f}%")
    gbs = tot_b / (tot_us * 1e-6) / 1e9
    print(f"{'all kernels of the chain':46s} {'':14s} {tot_us:9.1f} {tot_b / 1e6:13.1f} {gbs:7.0f} {100 * gbs / PEAK:8.1f}%")


if __name__ == "__main__":
    main()
