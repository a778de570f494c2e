#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run10.sh': TAA's history taps out of an LDS window (MIFX_TAA_WINDOW) -- parity suites, then the A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_bloom_taa.py tests/test_gpu_chain.py tests/test_gpu_host_sequence.py tests/test_gpu_sharded.py -q -x 2>&1 | tail -6 | tee gpurun_out/r04_taa_window_tests.txt
bash tools/ab_env.sh gather:MIFX_TAA_WINDOW=0 window:MIFX_TAA_WINDOW=1 gather2:MIFX_TAA_WINDOW=0 window2:MIFX_TAA_WINDOW=1 2>&1 | grep -i "kernel (avg\|taa_kernel\|sum of"
cp gpurun_out/abenv_table.txt gpurun_out/r04_ab_taa_window.txt
