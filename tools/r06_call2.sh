#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out
quiet() { grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
MIFX_PARITY_MEASURE=1 MIFX_PARITY_LOG=/tmp/par.jsonl timeout 900 python -m pytest tests/test_gpu_chain.py -k "full_size_parity or 3840x2160 or 8k_frame_pair" -q -s 2>&1 | quiet | grep -E "^3840|^7680|passed|failed|rror" > gpurun_out/r06_full_size_parity.txt
cat gpurun_out/r06_full_size_parity.txt
for v in plain mixed; do
  [ $v = mixed ] && export MIFX_A3_COPY_ROLE=6 || unset MIFX_A3_COPY_ROLE
  bash tools/gpu_run.sh pmc a3_${v}_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum > /dev/null
  bash tools/gpu_run.sh pmc a3_${v}_sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR > /dev/null
  grep -E "^kernel|ssao_compute_ao" gpurun_out/pmc_a3_${v}_tcp.txt gpurun_out/pmc_a3_${v}_sq.txt | cut -c1-330
done
