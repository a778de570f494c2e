#!/bin/bash
# A/B of libmifx.so build variants on the GPU box: for every diligentfx_amd/variants/<name>.so, per-kernel average durations
# (40 timed frames after 20 of warm-up: a 10-frame run measures the clock ramp -- in round 3 it showed -9 % for a change that is worth nothing at the steady-state clock)
# (rocprofv3 --kernel-trace --stats over a short bench run) and, with TESTS=1, the GPU parity suite.
#   gpurun -- 'TESTS=1 bash tools/ab_gpu.sh [name ...]'      -> gpurun_out/ab_<name>.txt, gpurun_out/ab_table.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
export MIFX_CHAIN_OVERLAP=0   # per-kernel durations: no concurrent streams
names=("$@")
if [ ${#names[@]} -eq 0 ]; then for f in diligentfx_amd/variants/*.so; do names+=("$(basename "$f" .so)"); done; fi
cp diligentfx_amd/libmifx.so /tmp/libmifx_orig.so
for n in "${names[@]}"; do
    cp "diligentfx_amd/variants/$n.so" diligentfx_amd/libmifx.so
    (cd /tmp && timeout ${STEP_TIMEOUT:-120} rocprofv3 --kernel-trace --stats -d "/tmp/ab_$n" -- python "$R/bench.py" --overlap 0 --steps 40 --warmup 20 --exact-warmup --no-overlap-check --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep --no-stage-lines > "/tmp/ab_$n.log" 2>&1)
    python tools/kernel_stats.py "/tmp/ab_$n" "$n" > "gpurun_out/ab_$n.txt" 2>&1
    if [ -n "$TESTS" ]; then timeout ${STEP_TIMEOUT:-120} python -m pytest tests -m gpu -q 2>&1 | tail -${TAIL:-15} > "gpurun_out/ab_${n}_tests.txt"; fi
done
cp /tmp/libmifx_orig.so diligentfx_amd/libmifx.so
python - "${names[@]}" <<'PY'
import sys
names = sys.argv[1:]
tab, order = {}, []
for n in names:
    for line in open(f"gpurun_out/ab_{n}.txt").read().splitlines()[2:]:
        f = line.split()
        if len(f) < 3:
            continue
        try:
            float(f[-1]); int(f[-3])
            k, avg, calls = " ".join(f[:-3]), float(f[-1]), int(f[-3])
        except ValueError:
            k, avg, calls = " ".join(f[:-1]), float(f[-1]), 0
        if k not in tab:
            tab[k] = {}
            order.append(k)
        tab[k][n] = (avg, calls)
with open("gpurun_out/ab_table.txt", "w") as out:
    out.write(f"{'kernel (avg us per launch)':52s} " + " ".join(f"{n[:12]:>12s}" for n in names) + "\n")
    for k in order:
        out.write(f"{k.replace('mifx::', '')[:52]:52s} " + " ".join(f"{tab[k].get(n, (float('nan'), 0))[0]:12.1f}" for n in names) + "\n")
print(open("gpurun_out/ab_table.txt").read())
PY
if [ -n "$TESTS" ]; then for n in "${names[@]}"; do echo "== tests $n"; tail -${TAIL:-15} "gpurun_out/ab_${n}_tests.txt"; done; fi
