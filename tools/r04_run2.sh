#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run2.sh': A3 with all 18 taps issued as one group (MIFX_A3_BATCH) at 4 / 5 / 6 waves per SIMD against the per-slice loop
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_gpu.sh a3base a3batch4 a3batch5 a3batch6 2>&1 | tail -40
cp gpurun_out/ab_table.txt gpurun_out/r04_ab_a3_batch.txt
cp diligentfx_amd/libmifx.so /tmp/orig.so
cp diligentfx_amd/variants/a3batch5.so diligentfx_amd/libmifx.so
timeout 300 python -m pytest tests/test_gpu_ssao.py tests/test_gpu_attribute_sweeps.py -q 2>&1 | tail -5 | tee gpurun_out/r04_a3batch5_tests.txt
cp /tmp/orig.so diligentfx_amd/libmifx.so
