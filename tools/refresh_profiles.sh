#!/bin/bash
# Re-measures what profiles/ holds for the current build on the GPU box:  gpurun -- 'bash tools/refresh_profiles.sh v7'
#   gpurun_out/kernel_stats_<tag>.txt   rocprofv3 --kernel-trace --stats of a short bench run (per-kernel average durations)
#   gpurun_out/pmc_fetch_size.txt, pmc_write_size.txt, pmc_traffic.json   two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) folded by tools/pmc_traffic.py
#   gpurun_out/bench_<tag>.json         the default bench line (roofline + cpu_baseline)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-now}
cd "$R" || exit 1
export TMPDIR=/tmp MIFX_CHAIN_OVERLAP=0
mkdir -p gpurun_out
ST=${STORAGE:+--storage $STORAGE}   # STORAGE=h4: the RGBA16_FLOAT storage build
B="python $R/bench.py --overlap 0 --steps 3 --warmup 4 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep $ST"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ks -- python "$R/bench.py" --overlap 0 --steps 40 --warmup 20 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep $ST > /tmp/ks.log 2>&1)
python tools/kernel_stats.py /tmp/ks "round 2 $tag, 3840x2160, 60 frames" > "gpurun_out/kernel_stats_$tag.txt" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 150 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- $B > /tmp/pmc_$c.log 2>&1)
    python tools/pmc_stats.py /tmp/pmc_$c $c > "gpurun_out/pmc_$(echo $c | tr 'A-Z' 'a-z').txt" 2>&1
done
python tools/pmc_traffic.py gpurun_out/pmc_fetch_size.txt gpurun_out/pmc_write_size.txt 7 "$tag" ${STORAGE:-fp32} > gpurun_out/pmc_traffic.json 2> gpurun_out/pmc_traffic.err
timeout 200 python bench.py $ST > "gpurun_out/bench_$tag.json" 2> "gpurun_out/bench_$tag.err"
tail -4 "gpurun_out/kernel_stats_$tag.txt"; head -c 600 gpurun_out/pmc_traffic.json; echo; head -c 400 "gpurun_out/bench_$tag.json"; echo
