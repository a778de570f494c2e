#!/bin/bash
# gpurun --timeout 700 -- 'bash tools/r03_run3.sh <tag>': SSR / chain parity tests, kernel stats, the VALU issue-rate microbenchmark, two SQ counter passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v3}
cd "$R" || exit 1
export TMPDIR=/tmp MIFX_CHAIN_OVERLAP=0
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ssr.py tests/test_gpu_steady_state.py tests/test_gpu_chain.py tests/test_gpu_attribute_sweeps.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -30 > "gpurun_out/r03_quick_tests_$tag.txt"
tail -4 "gpurun_out/r03_quick_tests_$tag.txt"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/ks -- python "$R/bench.py" --steps 40 --warmup 20 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep > /tmp/ks.log 2>&1)
python tools/kernel_stats.py /tmp/ks "round 3 $tag, 3840x2160, 60 frames" > "gpurun_out/r03_kernel_stats_$tag.txt" 2>&1
head -14 "gpurun_out/r03_kernel_stats_$tag.txt"
(cd tools/microbench && timeout 120 ./valu_rate3 50 > "$R/gpurun_out/r03_valu_issue_rate_$tag.txt" 2>&1); head -12 "gpurun_out/r03_valu_issue_rate_$tag.txt"
B="python $R/bench.py --steps 3 --warmup 4 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep"
(cd /tmp && timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d /tmp/pmc_sq1 -- $B > /tmp/pmc_sq1.log 2>&1)
python tools/pmc_stats.py /tmp/pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY > "gpurun_out/r03_pmc_sq_counters_$tag.txt" 2>&1
(cd /tmp && timeout 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_sq2 -- $B > /tmp/pmc_sq2.log 2>&1)
python tools/pmc_stats.py /tmp/pmc_sq2 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES > "gpurun_out/r03_pmc_sq_waits_$tag.txt" 2>&1
tail -3 /tmp/pmc_sq2.log | cut -c1-200
head -8 "gpurun_out/r03_pmc_sq_counters_$tag.txt"; head -8 "gpurun_out/r03_pmc_sq_waits_$tag.txt"
