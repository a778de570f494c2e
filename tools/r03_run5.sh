#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp MIFX_CHAIN_OVERLAP=0
mkdir -p gpurun_out
STEP_TIMEOUT=150 bash tools/ab_gpu.sh "$@" 2>&1 | tail -12
timeout 250 python bench.py > gpurun_out/r03_bench_v5.json 2> gpurun_out/r03_bench_v5.err; cut -c1-300 gpurun_out/r03_bench_v5.json; tail -3 gpurun_out/r03_bench_v5.err
