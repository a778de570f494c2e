#!/bin/bash
# gpurun --timeout 1200 -- 'bash tools/r04_run14.sh': band times + class row counts for three cut sets at 8K / 8 ranks (data for refitting tiling.cost_weighted_cuts)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/shard_cost.py --weighted --classes --ranks 0 1 2 3 4 5 6 7 --steps 8 2>&1 | grep -v amdgpu.ids > gpurun_out/fit_weighted.txt
W=$(grep "whole frame" gpurun_out/fit_weighted.txt | sed 's/.*whole frame \([0-9.]*\) ms.*/\1/')
timeout 300 python tools/shard_cost.py --classes --ranks 0 1 2 3 4 5 6 7 --steps 8 --whole-ms $W 2>&1 | grep -v amdgpu.ids > gpurun_out/fit_equal.txt
timeout 300 python tools/shard_cost.py --classes --ranks 0 1 2 3 4 5 6 7 --steps 8 --whole-ms $W --cuts 0 900 1500 1950 2350 2800 3250 3750 4320 2>&1 | grep -v amdgpu.ids > gpurun_out/fit_other.txt
grep -h "CLASSES\|whole frame\|slowest" gpurun_out/fit_weighted.txt gpurun_out/fit_equal.txt gpurun_out/fit_other.txt
