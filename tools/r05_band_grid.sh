#!/bin/bash
# gpurun --timeout 600 -- 'bash tools/r05_band_grid.sh [rank]': per-kernel times of ONE band of the 8K / 8-rank split (phases back to back on one stream), by launch grid
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rank=${1:-4}
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/grid$rank -- python "$R/tools/shard_cost.py" --weighted --overlap 0 --ranks $rank --whole-ms 7.0 --steps 12 > /tmp/grid$rank.log 2>&1)
tail -3 /tmp/grid$rank.log
python tools/grid_stats.py /tmp/grid$rank 6 > gpurun_out/r05_shard_grid_8k_rank$rank.txt 2>&1
cat gpurun_out/r05_shard_grid_8k_rank$rank.txt
