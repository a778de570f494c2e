#!/bin/bash
# gpurun --timeout 1500 -- 'bash tools/r03_run10.sh v9': everything profiles/ holds for the final build of round 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v9}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > "gpurun_out/r03_gpu_tests_$tag.txt"; grep -E "passed|failed" "gpurun_out/r03_gpu_tests_$tag.txt"
# the default command (two streams across frames) with its rocprofv3 summary, then the same build with one stream
timeout 300 python bench.py > "gpurun_out/r03_bench_$tag.json" 2> "gpurun_out/r03_bench_$tag.err"; head -c 300 "gpurun_out/r03_bench_$tag.json"; echo
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ksd -- python "$R/bench.py" --no-cpu-baseline > /tmp/ksd.log 2>&1)
python tools/kernel_stats.py /tmp/ksd "round 3 $tag, 3840x2160, python bench.py (two streams across frames; warm-up, sweep and per-stage frames included)" > "gpurun_out/r03_kernel_stats_${tag}_default_cmd.txt" 2>&1
python tools/overlap_stats.py /tmp/ksd > "gpurun_out/r03_overlap_stats_$tag.txt" 2>&1
export MIFX_CHAIN_OVERLAP=0
B="python $R/bench.py --overlap 0 --steps 3 --warmup 4 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ks -- python "$R/bench.py" --overlap 0 --steps 40 --warmup 20 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep > /tmp/ks.log 2>&1)
python tools/kernel_stats.py /tmp/ks "round 3 $tag, 3840x2160, 60 frames, one stream" > "gpurun_out/r03_kernel_stats_$tag.txt" 2>&1; head -8 "gpurun_out/r03_kernel_stats_$tag.txt"
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 150 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- $B > /tmp/pmc_$c.log 2>&1)
    python tools/pmc_stats.py /tmp/pmc_$c $c > "gpurun_out/r03_pmc_$(echo $c | tr 'A-Z' 'a-z')_$tag.txt" 2>&1
done
python tools/pmc_traffic.py "gpurun_out/r03_pmc_fetch_size_$tag.txt" "gpurun_out/r03_pmc_write_size_$tag.txt" 7 "$tag" fp32 > gpurun_out/r03_pmc_traffic.json 2> gpurun_out/r03_pmc_traffic.err; head -c 500 gpurun_out/r03_pmc_traffic.json; echo
(cd /tmp && timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d /tmp/pmc_sq1 -- $B > /tmp/pmc_sq1.log 2>&1)
python tools/pmc_stats.py /tmp/pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY > "gpurun_out/r03_pmc_sq_counters_$tag.txt" 2>&1
(cd /tmp && timeout 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_sq2 -- $B > /tmp/pmc_sq2.log 2>&1)
python tools/pmc_stats.py /tmp/pmc_sq2 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES > "gpurun_out/r03_pmc_sq_waits_$tag.txt" 2>&1
(cd /tmp && timeout 100 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d /tmp/fg -- "$R/tools/microbench/fetch_granularity" > /tmp/fg.log 2>&1)
(cat /tmp/fg.log | grep stride; python tools/pmc_stats.py /tmp/fg TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum) > "gpurun_out/r03_fetch_granularity.txt" 2>&1; cat gpurun_out/r03_fetch_granularity.txt
(cd /tmp && timeout 100 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d /tmp/rq -- $B > /tmp/rq.log 2>&1)
python tools/pmc_stats.py /tmp/rq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum > "gpurun_out/r03_pmc_rdreq_$tag.txt" 2>&1; head -6 "gpurun_out/r03_pmc_rdreq_$tag.txt"
unset MIFX_CHAIN_OVERLAP
timeout 200 python bench.py --overlap 0 > "gpurun_out/r03_bench_${tag}_one_stream.json" 2>/dev/null
timeout 200 python bench.py --config ssao1080 > "gpurun_out/r03_bench_ssao1080_$tag.json" 2>/dev/null
timeout 200 python bench.py --config pbr4k > "gpurun_out/r03_bench_pbr4k_$tag.json" 2>/dev/null
timeout 200 python bench.py --storage h4 --no-cpu-baseline > "gpurun_out/r03_bench_h4_$tag.json" 2>/dev/null
for f in r03_bench_$tag r03_bench_${tag}_one_stream r03_bench_ssao1080_$tag r03_bench_pbr4k_$tag r03_bench_h4_$tag; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(f"gpurun_out/{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], j["ms_per_step"], j["value"], j.get("roofline", {}).get("whole_chain", {}).get("frac"))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
done
