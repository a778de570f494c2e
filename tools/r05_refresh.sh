#!/bin/bash
# gpurun --timeout 1500 -- 'bash tools/r05_refresh.sh v14': everything profiles/ holds for the build that ships -- kernel stats (one stream, 60 frames), the default command under
# [round 5's first GPU call: the state round 4 left, re-measured on this round's box, + the full GPU suite and the layered shade's timing]
# rocprofv3 (three lanes: what runs beside what), the two PMC passes folded into HBM bytes per stage, TCP look-ups per kernel, the default bench line, the configs[1] / [2] lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v1}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ks -- python "$R/bench.py" --overlap 0 --steps 40 --warmup 20 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep --no-stage-lines > /tmp/ks.log 2>&1)
python tools/kernel_stats.py /tmp/ks "round 5 $tag, 3840x2160, 60 frames, one stream" > "gpurun_out/r05_kernel_stats_$tag.txt" 2>&1; head -12 "gpurun_out/r05_kernel_stats_$tag.txt"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ksd -- python "$R/bench.py" --no-cpu-baseline --no-stage-lines > /tmp/ksd.log 2>&1)
python tools/kernel_stats.py /tmp/ksd "round 5 $tag, 3840x2160, python bench.py (three lanes across frames; warm-up, sweep and per-stage frames included)" > "gpurun_out/r05_kernel_stats_${tag}_default_cmd.txt" 2>&1
python tools/overlap_stats.py /tmp/ksd > "gpurun_out/r05_overlap_stats_$tag.txt" 2>&1
B="python $R/bench.py --overlap 0 --steps 3 --warmup 4 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep --no-stage-lines"
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && MIFX_CHAIN_OVERLAP=0 timeout 150 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- $B > /tmp/pmc_$c.log 2>&1)
    python tools/pmc_stats.py /tmp/pmc_$c $c > "gpurun_out/r05_pmc_$(echo $c | tr 'A-Z' 'a-z')_$tag.txt" 2>&1
done
python tools/pmc_traffic.py "gpurun_out/r05_pmc_fetch_size_$tag.txt" "gpurun_out/r05_pmc_write_size_$tag.txt" 7 "$tag" fp32 > gpurun_out/r05_pmc_traffic.json 2> gpurun_out/r05_pmc_traffic.err
(cd /tmp && MIFX_CHAIN_OVERLAP=0 timeout 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace -d /tmp/pmc_tcp -- $B > /tmp/pmc_tcp.log 2>&1)
python tools/pmc_stats.py /tmp/pmc_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum > "gpurun_out/r05_pmc_tcp_$tag.txt" 2>&1
cp gpurun_out/r05_pmc_traffic.json profiles/r05_pmc_traffic.json   # (bench.py reads the latest committed measurement: this run's own)
timeout 300 python bench.py > "gpurun_out/r05_bench_$tag.json" 2> "gpurun_out/r05_bench_$tag.err"
timeout 200 python bench.py --overlap 0 --no-cpu-baseline --no-stage-lines > "gpurun_out/r05_bench_${tag}_one_stream.json" 2>/dev/null
timeout 200 python bench.py --config ssao1080 > "gpurun_out/r05_bench_ssao1080_$tag.json" 2>/dev/null
timeout 200 python bench.py --config pbr4k > "gpurun_out/r05_bench_pbr4k_$tag.json" 2>/dev/null
python - "$tag" <<'PY'
import json, sys
t = sys.argv[1]
for n in (f"r05_bench_{t}", f"r05_bench_{t}_one_stream", f"r05_bench_ssao1080_{t}", f"r05_bench_pbr4k_{t}"):
    try:
        d = json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"], d.get("roofline", {}).get("frac"), d["config"].get("chain_hbm_frac"), d.get("cpu_baseline", {}).get("value"), json.dumps(d["config"].get("stage_lines"))[:300])
    except Exception as e:
        print(n, "FAILED", e)
PY
timeout 120 python tools/layers_timing.py --ab --steps 12 > "gpurun_out/r05_layers_timing_$tag.txt" 2>&1; grep -v amdgpu.ids "gpurun_out/r05_layers_timing_$tag.txt" | tail -22
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5; python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > "gpurun_out/r05_gpu_tests_$tag.txt"; tail -4 "gpurun_out/r05_gpu_tests_$tag.txt"
head -c 500 gpurun_out/r05_pmc_traffic.json; echo; head -8 "gpurun_out/r05_overlap_stats_$tag.txt"
