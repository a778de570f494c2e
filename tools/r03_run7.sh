#!/bin/bash
# full GPU suite + kernel stats + the three bench lines (chain, ssao1080, pbr4k)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v7}
cd "$R" || exit 1
export TMPDIR=/tmp MIFX_CHAIN_OVERLAP=0
mkdir -p gpurun_out
timeout ${TEST_TIMEOUT:-500} python -m pytest tests -m gpu -q ${PYTEST_ARGS} 2>&1 | tail -40 > "gpurun_out/r03_gpu_tests_$tag.txt"
tail -5 "gpurun_out/r03_gpu_tests_$tag.txt"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/ks -- python "$R/bench.py" --steps 40 --warmup 20 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep > /tmp/ks.log 2>&1)
python tools/kernel_stats.py /tmp/ks "round 3 $tag, 3840x2160, 60 frames" > "gpurun_out/r03_kernel_stats_$tag.txt" 2>&1
head -30 "gpurun_out/r03_kernel_stats_$tag.txt"
timeout 250 python bench.py > "gpurun_out/r03_bench_$tag.json" 2> "gpurun_out/r03_bench_$tag.err"; cut -c1-260 "gpurun_out/r03_bench_$tag.json"; tail -2 "gpurun_out/r03_bench_$tag.err"
if [ -z "$SKIP_STAGE" ]; then
for c in ssao1080 pbr4k; do timeout 200 python bench.py --config $c > "gpurun_out/r03_bench_${c}_$tag.json" 2> "gpurun_out/r03_bench_${c}_$tag.err"; cut -c1-260 "gpurun_out/r03_bench_${c}_$tag.json"; tail -2 "gpurun_out/r03_bench_${c}_$tag.err"; done
fi
