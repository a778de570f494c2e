#!/bin/bash
# A/B of run-time switches on the GPU box (the environment knobs of libmifx.so): per-kernel average durations of one build under several environments, one stream,
# 40 timed frames after 20 of warm-up (like tools/ab_gpu.sh, which compares builds).
#   gpurun -- 'bash tools/ab_env.sh base: win:MIFX_R5_WINDOW=1 ...'   ("name:VAR=VAL,VAR=VAL"; an empty list = the default environment) -> gpurun_out/abenv_table.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
export MIFX_CHAIN_OVERLAP=0
mkdir -p gpurun_out
names=()
for spec in "$@"; do
    n=${spec%%:*}; e=${spec#*:}
    names+=("$n")
    envs=()
    IFS=',' read -ra kv <<< "$e"
    for x in "${kv[@]}"; do [ -n "$x" ] && envs+=("$x"); done
    (cd /tmp && env "${envs[@]}" timeout ${STEP_TIMEOUT:-120} rocprofv3 --kernel-trace --stats -d "/tmp/abenv_$n" -- python "$R/bench.py" --overlap 0 --steps 40 --warmup 20 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep --no-stage-lines > "/tmp/abenv_$n.log" 2>&1)
    python tools/kernel_stats.py "/tmp/abenv_$n" "$n ($e)" > "gpurun_out/abenv_$n.txt" 2>&1
done
python - "${names[@]}" <<'PY'
import sys
names = sys.argv[1:]
tab, order = {}, []
for n in names:
    for line in open(f"gpurun_out/abenv_{n}.txt").read().splitlines()[2:]:
        f = line.split()
        if len(f) < 3:
            continue
        try:
            float(f[-1]); int(f[-3])
            k, avg = " ".join(f[:-3]), float(f[-1])
        except ValueError:
            k, avg = " ".join(f[:-1]), float(f[-1])
        if k not in tab:
            tab[k] = {}
            order.append(k)
        tab[k][n] = avg
with open("gpurun_out/abenv_table.txt", "w") as out:
    out.write(f"{'kernel (avg us per launch)':52s} " + " ".join(f"{n[:12]:>12s}" for n in names) + "\n")
    for k in order:
        out.write(f"{k.replace('mifx::', '')[:52]:52s} " + " ".join(f"{tab[k].get(n, float('nan')):12.1f}" for n in names) + "\n")
print(open("gpurun_out/abenv_table.txt").read())
PY
