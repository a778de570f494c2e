#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run3.sh': R5 with its sixteen taps out of an LDS window against the gather form; the SSR / sharding / chain parity tests on the new kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ssr.py tests/test_gpu_sharded.py tests/test_gpu_chain.py -q -x 2>&1 | tail -6 | tee gpurun_out/r04_r5_window_tests.txt
bash tools/ab_env.sh gather:MIFX_R5_WINDOW=0 window:MIFX_R5_WINDOW=1 gather2:MIFX_R5_WINDOW=0 window2:MIFX_R5_WINDOW=1
cp gpurun_out/abenv_table.txt gpurun_out/r04_ab_r5_window.txt
