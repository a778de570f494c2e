#!/bin/bash
# gpurun --timeout 600 -- 'bash tools/r03_run2.sh <tag>': selected tests first (fast feedback), the suite, stats of the resolve's paths, A/B kernel stats, bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v2}
cd "$R" || exit 1
export TMPDIR=/tmp MIFX_CHAIN_OVERLAP=0
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ssao.py tests/test_gpu_chain.py::test_chain_fusion_is_bit_identical tests/test_gpu_storage_h4.py tests/test_gpu_ssr.py -m gpu -q -x 2>&1 | tail -30 > "gpurun_out/r03_quick_tests_$tag.txt"
tail -4 "gpurun_out/r03_quick_tests_$tag.txt"
timeout 120 python tools/ssao_stats.py > "gpurun_out/r03_ssao_paths_$tag.txt" 2>&1; tail -8 "gpurun_out/r03_ssao_paths_$tag.txt"
for m in ${MASKS:-15 3}; do
    (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d "/tmp/ks_$m" -- python "$R/bench.py" --steps 40 --warmup 20 --fusion-mask $m --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep > "/tmp/ks_$m.log" 2>&1)
    python tools/kernel_stats.py "/tmp/ks_$m" "round 3 $tag, fusion mask $m, 3840x2160, 60 frames" > "gpurun_out/r03_kernel_stats_${tag}_mask$m.txt" 2>&1
    grep -i "error\|Traceback" "/tmp/ks_$m.log" | head -5
done
head -12 "gpurun_out/r03_kernel_stats_${tag}_mask15.txt"
if [ -z "$SKIP_SUITE" ]; then
    timeout ${TEST_TIMEOUT:-400} python -m pytest tests -m gpu -q ${PYTEST_ARGS} 2>&1 | tail -40 > "gpurun_out/r03_gpu_tests_$tag.txt"
    tail -5 "gpurun_out/r03_gpu_tests_$tag.txt"
fi
timeout 250 python bench.py > "gpurun_out/r03_bench_$tag.json" 2> "gpurun_out/r03_bench_$tag.err"
cut -c1-400 "gpurun_out/r03_bench_$tag.json"
