#!/bin/bash
# gpurun --timeout 600 -- 'bash tools/r03_run8.sh <tag>': texture-addresser / L1 / L2 counter passes (what do the latency-bound kernels wait for?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v8}
cd "$R" || exit 1
export TMPDIR=/tmp MIFX_CHAIN_OVERLAP=0
mkdir -p gpurun_out
B="python $R/bench.py --steps 3 --warmup 4 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep"
pass() { # name counters...
    n=$1; shift
    (cd /tmp && timeout 120 rocprofv3 --pmc "$@" --kernel-trace -d "/tmp/pmc_$n" -- $B > "/tmp/pmc_$n.log" 2>&1) || { echo "pass $n failed"; tail -5 "/tmp/pmc_$n.log" | cut -c1-300; }
    python tools/pmc_stats.py "/tmp/pmc_$n" "$@" > "gpurun_out/r03_pmc_${n}_$tag.txt" 2>&1
    head -9 "gpurun_out/r03_pmc_${n}_$tag.txt" | cut -c1-260
}
pass ta1 TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN2_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
