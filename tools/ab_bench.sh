#!/bin/bash
# A/B of whole-frame configurations of ONE build on the GPU box: the default bench loop (no CPU baseline, no sweeps) per configuration, interleaved and repeated, frame
# time as mean and median.  What tools/ab_gpu.sh (builds) and tools/ab_env.sh (environment knobs, per-kernel times on one stream) do not cover: the multi-stream modes.
#   gpurun -- 'REPS=2 bash tools/ab_bench.sh "ov3|--overlap 3" "ov4|--overlap 4" "ov4e|--overlap 4 --lane-edges a<b@1" "cap|MIFX_X=1 --overlap 3"'
# ("name|[VAR=VALUE ...] bench arguments": leading VAR=VALUE words go to the environment)  -> gpurun_out/ab_bench.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/${OUT:-ab_bench.txt}
: > "$out"
for rep in $(seq 1 ${REPS:-2}); do
    for spec in "$@"; do
        n=${spec%%|*}; a=${spec#*|}
        envs=(); rest=()
        for wd in $a; do
            if [ ${#rest[@]} -eq 0 ] && [[ "$wd" == *=* ]] && [[ "$wd" != --* ]]; then envs+=("$wd"); else rest+=("$wd"); fi
        done
        line=$(env "${envs[@]}" timeout ${STEP_TIMEOUT:-150} python bench.py --steps ${STEPS:-60} --warmup ${WARMUP:-24} --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep --no-stage-lines "${rest[@]}" 2>/tmp/ab_bench_err.txt | tail -1)
        python - "$n" "$rep" "$line" <<'PY' | tee -a "$out"
import json, sys
n, rep, line = sys.argv[1:4]
try:
    d = json.loads(line)
    print(f"{n:28s} rep {rep}  ms_per_step {d['ms_per_step']:.4f}  median {d['ms_per_step_median']:.4f}")
except Exception as e:
    print(f"{n:28s} rep {rep}  FAILED {e!r}: {line[:200]!r} {open('/tmp/ab_bench_err.txt').read()[-600:]!r}")
PY
    done
done
