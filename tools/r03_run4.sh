#!/bin/bash
# gpurun --timeout 800 -- 'bash tools/r03_run4.sh': parity tests of the working-tree library, then the A/B of the build variants in diligentfx_amd/variants/
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp MIFX_CHAIN_OVERLAP=0
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_ssr.py tests/test_gpu_ssao.py tests/test_gpu_bloom_taa.py tests/test_gpu_steady_state.py tests/test_gpu_chain.py tests/test_gpu_attribute_sweeps.py tests/test_gpu_sharded.py tests/test_gpu_storage_h4.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r03_quick_tests_v4.txt
tail -4 gpurun_out/r03_quick_tests_v4.txt
STEP_TIMEOUT=150 bash tools/ab_gpu.sh "$@" 2>&1 | tail -45
