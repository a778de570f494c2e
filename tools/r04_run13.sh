#!/bin/bash
# gpurun --timeout 600 -- 'bash tools/r04_run13.sh': per-kernel times of ONE interior band of the 8K / 8-rank split (rank 4), band launches beside the whole-frame launches
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/grid4 -- python "$R/tools/shard_cost.py" --weighted --ranks 4 --whole-ms 7.3 --steps 12 > /tmp/grid4.log 2>&1)
tail -3 /tmp/grid4.log
python tools/grid_stats.py /tmp/grid4 6 > gpurun_out/r04_shard_grid_8k_rank4.txt 2>&1
tail -45 gpurun_out/r04_shard_grid_8k_rank4.txt
