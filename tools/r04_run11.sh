#!/bin/bash
# gpurun --timeout 1200 -- 'bash tools/r04_run11.sh': the sharded path with phases 0 and 1 on two streams: parity suites, then the compute-side cost at 8K / 8 ranks with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_sharded.py tests/test_comm.py -q -x 2>&1 | tail -4
MIFX_SHARD_OVERLAP=0 timeout 400 python tools/shard_cost.py --weighted --ranks 0 1 2 3 4 5 6 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_shard_cost_8k_weighted_serial.txt | tail -4
timeout 400 python tools/shard_cost.py --weighted --ranks 0 1 2 3 4 5 6 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_shard_cost_8k_weighted.txt | tail -12
