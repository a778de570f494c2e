#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run8.sh': the product's host objects on the scenarios pinned to the executed reference host (tests/test_gpu_host_sequence.py), then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_host_sequence.py -q -s 2>&1 | tail -25 | tee gpurun_out/r04_gpu_host_sequence.txt
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r04_gpu_tests_v14.txt
