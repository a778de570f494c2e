#!/bin/bash
# The GPU-box command sequences of this repository as ONE parametrised script (rounds 3 and 4 kept a one-off script per call: git history has them).
#   gpurun --timeout <s> -- 'bash tools/gpu_run.sh <recipe> [arguments]'          everything lands in gpurun_out/ (copy what is to be kept into profiles/)
#
#   tests [pytest arguments]        the GPU suite (default: all of it, -m gpu) + smoke(); the summary line is what is printed
#   refresh <tag>                   what profiles/ holds for the build that ships: kernel stats (one stream, 60 frames), the default command under rocprofv3 + what runs
#                                   beside what, the PMC passes (FETCH_SIZE, WRITE_SIZE -> HBM bytes per kernel; TCP look-ups; SQ counters), the bench lines
#   pmc <name> <COUNTER> ...        one counter pass over three one-stream frames -> gpurun_out/pmc_<name>.txt   (--pmc runs alone: no other trace domains)
#   parity-table                    the outlier fraction of every comparison of the GPU suite, shipped build beside diligentfx_amd/variants/strict.so
#   band-grid [rank]                per-kernel times of one band of the 8K / 8-rank split, by launch grid (phases back to back on one stream)
#   bench-procs [n [w h]]           bench.py --gpus n as n processes on this one GPU through the library's RCCL branch (stand-in transport): self test, calibration, verification
#   shard-cost [arguments]          tools/shard_cost.py --weighted --refine 2 [arguments] (compute-side cost of the sharded frame; --overlap 0 | 2)
#   ab-builds | ab-env | ab-bench   tools/ab_gpu.sh | ab_env.sh | ab_bench.sh with the remaining arguments
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
recipe=${1:-tests}; shift
quiet() { grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
BENCH1="python $R/bench.py --overlap 0 --steps 3 --warmup 4 --exact-warmup --no-overlap-check --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep --no-stage-lines"
pmc_pass() { # name counters...
    local n=$1; shift
    (cd /tmp && MIFX_CHAIN_OVERLAP=0 timeout 150 rocprofv3 --pmc "$@" --kernel-trace -d "/tmp/pmc_$n" -- $BENCH1 > "/tmp/pmc_$n.log" 2>&1) || { echo "pass $n failed"; tail -5 "/tmp/pmc_$n.log" | cut -c1-300; }
    python tools/pmc_stats.py "/tmp/pmc_$n" "$@" > "gpurun_out/pmc_$n.txt" 2>&1
}
case "$recipe" in
tests)
    if [ $# -eq 0 ]; then set -- tests -m gpu; fi
    timeout ${TEST_TIMEOUT:-900} python -m pytest "$@" -q 2>&1 | quiet > gpurun_out/gpu_tests_full.txt
    grep -E "passed|failed|error" gpurun_out/gpu_tests_full.txt | tail -3 | tee "gpurun_out/${TAG:-r06}_gpu_tests.txt"
    timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | quiet | tail -2 | tee -a "gpurun_out/${TAG:-r06}_gpu_tests.txt"
    ;;
refresh)
    tag=${1:-v1}; rd=${ROUND:-r06}
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ks -- python "$R/bench.py" --overlap 0 --steps 40 --warmup 20 --exact-warmup --no-overlap-check --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep --no-stage-lines > /tmp/ks.log 2>&1)
    python tools/kernel_stats.py /tmp/ks "round ${rd#r0} $tag, 3840x2160, 60 frames, one stream" > "gpurun_out/${rd}_kernel_stats_$tag.txt" 2>&1; head -12 "gpurun_out/${rd}_kernel_stats_$tag.txt"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ksd -- python "$R/bench.py" --no-cpu-baseline --no-stage-lines > /tmp/ksd.log 2>&1)
    python tools/kernel_stats.py /tmp/ksd "round ${rd#r0} $tag, 3840x2160, python bench.py (three lanes across frames; warm-up, sweep and per-stage frames included)" > "gpurun_out/${rd}_kernel_stats_${tag}_default_cmd.txt" 2>&1
    python tools/overlap_stats.py /tmp/ksd > "gpurun_out/${rd}_overlap_stats_$tag.txt" 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        pmc_pass $c $c
        cp "gpurun_out/pmc_$c.txt" "gpurun_out/${rd}_pmc_$(echo $c | tr 'A-Z' 'a-z')_$tag.txt"
    done
    python tools/pmc_traffic.py "gpurun_out/${rd}_pmc_fetch_size_$tag.txt" "gpurun_out/${rd}_pmc_write_size_$tag.txt" 7 "$tag" fp32 > "gpurun_out/${rd}_pmc_traffic.json" 2> "gpurun_out/${rd}_pmc_traffic.err"
    pmc_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum; cp gpurun_out/pmc_tcp.txt "gpurun_out/${rd}_pmc_tcp_$tag.txt"
    pmc_pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY; cp gpurun_out/pmc_sq.txt "gpurun_out/${rd}_pmc_sq_counters_$tag.txt"
    # (bench.py reads the latest committed counter files: this run's own, so that the line's traffic / speed-of-light table belong to the build it times)
    cp "gpurun_out/${rd}_pmc_traffic.json" "gpurun_out/${rd}_pmc_tcp_$tag.txt" "gpurun_out/${rd}_pmc_sq_counters_$tag.txt" profiles/
    timeout 300 python bench.py > "gpurun_out/${rd}_bench_$tag.json" 2> "gpurun_out/${rd}_bench_$tag.err"
    timeout 200 python bench.py --overlap 0 --no-cpu-baseline --no-stage-lines > "gpurun_out/${rd}_bench_${tag}_one_stream.json" 2>/dev/null
    timeout 200 python bench.py --config ssao1080 > "gpurun_out/${rd}_bench_ssao1080_$tag.json" 2>/dev/null
    timeout 200 python bench.py --config pbr4k > "gpurun_out/${rd}_bench_pbr4k_$tag.json" 2>/dev/null
    python - "$rd" "$tag" <<'PY'
import json, sys
rd, t = sys.argv[1:3]
for n in (f"{rd}_bench_{t}", f"{rd}_bench_{t}_one_stream", f"{rd}_bench_ssao1080_{t}", f"{rd}_bench_pbr4k_{t}"):
    try:
        d = json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"], d.get("roofline", {}).get("frac"), d["config"].get("chain_hbm_frac"), d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(n, "FAILED", e)
PY
    head -8 "gpurun_out/${rd}_overlap_stats_$tag.txt"
    ;;
pmc)
    n=$1; shift
    pmc_pass "$n" "$@"; head -30 "gpurun_out/pmc_$n.txt" | cut -c1-260
    ;;
parity-table)
    rm -f /tmp/shipped.jsonl /tmp/strict.jsonl
    MIFX_PARITY_LOG=/tmp/shipped.jsonl timeout 600 python -m pytest tests -m gpu -q 2>&1 | quiet | tail -3
    MIFX_LIB_PATH=$R/diligentfx_amd/variants/strict.so MIFX_PARITY_LOG=/tmp/strict.jsonl MIFX_PARITY_MEASURE=1 timeout 600 python -m pytest tests -m gpu -q 2>&1 | quiet | tail -3
    python tools/parity_table.py /tmp/shipped.jsonl /tmp/strict.jsonl > gpurun_out/parity_outliers_strict_vs_shipped.txt
    wc -l gpurun_out/parity_outliers_strict_vs_shipped.txt
    ;;
band-grid)
    rank=${1:-4}
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/grid$rank -- python "$R/tools/shard_cost.py" --weighted --overlap 0 --ranks $rank --whole-ms 7.0 --steps 12 > /tmp/grid$rank.log 2>&1)
    tail -3 /tmp/grid$rank.log
    python tools/grid_stats.py /tmp/grid$rank 6 | tee "gpurun_out/shard_grid_8k_rank$rank.txt"
    ;;
shard-cost)
    timeout 500 python tools/shard_cost.py --weighted --refine 2 "$@" 2>&1 | quiet | tee gpurun_out/shard_cost_8k.txt | grep -E "whole|slowest|refined"
    ;;
bench-procs)
    # bench.py's N > 1 flow on ONE GPU: N processes (default 4) through the library's RCCL branch with the test-only stand-in transport (tests/fake_rccl), start-up self test,
    # band calibration, timed frames, shard verification -- everything of the driver's multi-GPU run except the real librccl and the other GPUs
    n=${1:-4}; bw=${2:-1920}; bh=${3:-1080}   # (3840 2160 = the driver's own N > 1 workload: one 7680x4320 frame)
    g++ -shared -fPIC -O1 -std=c++17 -w -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include tests/fake_rccl/fake_rccl.cpp -o /tmp/librccl_fake.so -L /opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib || exit 1
    # (round 6: the plain command -- bench.py starts its own ranks; the side channel is gloo, the rows travel inside libmifx)
    MIFX_RCCL_PATH=/tmp/librccl_fake.so HSA_ENABLE_IPC_MODE_LEGACY=0 timeout ${BENCH_TIMEOUT:-400} python bench.py --gpus $n --single-gpu --width $bw --height $bh --steps 6 --warmup 8 ${BENCH_EXTRA:-} \
        > "gpurun_out/${TAG:-r06}_bench_${n}procs_standin_transport.json" 2> /tmp/bp.err
    tail -c 600 /tmp/bp.err | quiet
    python - "gpurun_out/${TAG:-r06}_bench_${n}procs_standin_transport.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["n_gpus"], "processes:", d["ms_per_step"], "ms;", d["config"]["sharding"][-160:], "| verified:", d.get("shard_verified"), "| comm:", {k: v for k, v in d.get("comm", {}).items() if k != "exchange_ms"},
      "| same frame on one GPU:", d.get("single_gpu_same_frame_ms"))
PY
    ;;
ab-builds) bash tools/ab_gpu.sh "$@" ;;
ab-env) bash tools/ab_env.sh "$@" ;;
ab-bench) bash tools/ab_bench.sh "$@" ;;
*) echo "unknown recipe $recipe"; exit 2 ;;
esac
