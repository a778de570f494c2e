#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run1.sh': the three-lane mode (mifx_chain_set_overlap 3) -- bit-identity tests, frame time against mode 2, occupancy caps of the two
# gather kernels beside each other (MIFX_R4_LDS_PAD / MIFX_A3_LDS_PAD), and what runs beside what (overlap_stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_chain.py -q -k "overlap" 2>&1 | tail -5 > gpurun_out/r04_run1_tests.txt
cat gpurun_out/r04_run1_tests.txt
B="--steps 60 --warmup 30 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep"
run() { # name, env..., overlap
    local name=$1; shift
    local ov=$1; shift
    env "$@" timeout 200 python bench.py --overlap "$ov" $B 2>/tmp/err_$name.log | tail -1 > "/tmp/line_$name.json"
    python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"/tmp/line_{n}.json").read())
    print(f"{n:34s} ms_per_step {d['ms_per_step']:.4f}  median {d.get('ms_per_step_median')}")
except Exception as e:
    print(f"{n:34s} FAILED {e}: {open(f'/tmp/err_{n}.log').read()[-400:]}")
PY
}
{
run ov2 2 X=1
run ov3 3 X=1
run ov3_r4cap4 3 MIFX_R4_LDS_PAD=36864
run ov3_r4cap5 3 MIFX_R4_LDS_PAD=28672
run ov3_r4cap6 3 MIFX_R4_LDS_PAD=24576
run ov3_r4cap4_a3cap4 3 MIFX_R4_LDS_PAD=36864 MIFX_A3_LDS_PAD=36864
run ov3_a3cap4 3 MIFX_A3_LDS_PAD=36864
run ov2_again 2 X=1
run ov3_again 3 X=1
} 2>&1 | tee gpurun_out/r04_run1_ab.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks3 -- python "$R/bench.py" --overlap 3 $B > /tmp/ks3.log 2>&1)
python tools/overlap_stats.py /tmp/ks3 > gpurun_out/r04_overlap_stats_ov3.txt 2>&1
head -30 gpurun_out/r04_overlap_stats_ov3.txt
