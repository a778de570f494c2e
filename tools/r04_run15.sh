#!/bin/bash
# gpurun --timeout 600 -- 'bash tools/r04_run15.sh v17': the SQ counters of the shipped build (VALU instructions issued, wave cycles, waits), one stream: what DESIGN 4 "Round 4" rests
# its "bound twice over" statement on (VALU issue share = SQ_INSTS_VALU / SIMDs x ~3.1 clocks against the kernel's duration)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v17}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python $R/bench.py --overlap 0 --steps 3 --warmup 4 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep --no-stage-lines"
(cd /tmp && MIFX_CHAIN_OVERLAP=0 timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d /tmp/pmc_sq1 -- $B > /tmp/pmc_sq1.log 2>&1)
python tools/pmc_stats.py /tmp/pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY > "gpurun_out/r04_pmc_sq_counters_$tag.txt" 2>&1
head -14 "gpurun_out/r04_pmc_sq_counters_$tag.txt"
