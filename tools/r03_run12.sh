#!/bin/bash
# gpurun --timeout 600 -- 'bash tools/r03_run12.sh v12': kernel stats of the final build -- one stream (60 frames) and the default command
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v12}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ks -- python "$R/bench.py" --overlap 0 --steps 40 --warmup 20 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep > /tmp/ks.log 2>&1)
python tools/kernel_stats.py /tmp/ks "round 3 $tag, 3840x2160, 60 frames, one stream" > "gpurun_out/r03_kernel_stats_$tag.txt" 2>&1; head -12 "gpurun_out/r03_kernel_stats_$tag.txt"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ksd -- python "$R/bench.py" --no-cpu-baseline > /tmp/ksd.log 2>&1)
python tools/kernel_stats.py /tmp/ksd "round 3 $tag, 3840x2160, python bench.py (two streams across frames; warm-up, sweep and per-stage frames included)" > "gpurun_out/r03_kernel_stats_${tag}_default_cmd.txt" 2>&1
python tools/overlap_stats.py /tmp/ksd > "gpurun_out/r03_overlap_stats_$tag.txt" 2>&1
head -4 "gpurun_out/r03_kernel_stats_${tag}_default_cmd.txt"; tail -1 /tmp/ksd.log | cut -c1-200
