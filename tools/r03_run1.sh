#!/bin/bash
# Round-3 GPU call: the parity suite, then per-kernel averages of the chain with every fusion on against round 2's chain (fusion mask 3), then the bench line.
#   gpurun --timeout 600 -- 'bash tools/r03_run1.sh <tag>'
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v1}
cd "$R" || exit 1
export TMPDIR=/tmp MIFX_CHAIN_OVERLAP=0
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
    timeout ${TEST_TIMEOUT:-400} python -m pytest tests -m gpu -q -x ${PYTEST_ARGS} 2>&1 | tail -40 > "gpurun_out/r03_gpu_tests_$tag.txt"
    tail -5 "gpurun_out/r03_gpu_tests_$tag.txt"
fi
for m in ${MASKS:-15 3}; do
    (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d "/tmp/ks_$m" -- python "$R/bench.py" --steps 40 --warmup 20 --fusion-mask $m --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep > "/tmp/ks_$m.log" 2>&1)
    python tools/kernel_stats.py "/tmp/ks_$m" "round 3 $tag, fusion mask $m, 3840x2160, 60 frames" > "gpurun_out/r03_kernel_stats_${tag}_mask$m.txt" 2>&1
    tail -3 "/tmp/ks_$m.log" | cut -c1-300
done
timeout 250 python bench.py > "gpurun_out/r03_bench_$tag.json" 2> "gpurun_out/r03_bench_$tag.err"
cut -c1-700 "gpurun_out/r03_bench_$tag.json"
