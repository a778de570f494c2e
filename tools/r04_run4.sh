#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run4.sh': the Bloom output on demand (MIFX_CHAIN_FUSE_BLOOM_OUTPUT_ON_DEMAND): parity suites that read the plane, frame time with / without
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_storage_h4.py tests/test_gpu_dof.py tests/test_gpu_sharded.py tests/test_gpu_autoexposure.py tests/test_gpu_bloom_taa.py -q -x 2>&1 | tail -6 | tee gpurun_out/r04_bloom_on_demand_tests.txt
B="--steps 60 --warmup 30 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep"
for spec in "m15_ov2:--fusion-mask 15 --overlap 2" "m31_ov2:--fusion-mask 31 --overlap 2" "m15_ov3:--fusion-mask 15 --overlap 3" "m31_ov3:--fusion-mask 31 --overlap 3" "m15_ov0:--fusion-mask 15 --overlap 0" "m31_ov0:--fusion-mask 31 --overlap 0" "m15_ov2b:--fusion-mask 15 --overlap 2" "m31_ov2b:--fusion-mask 31 --overlap 2"; do
    n=${spec%%:*}; a=${spec#*:}
    timeout 200 python bench.py $a $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], d['ms_per_step_median'])"
done | tee gpurun_out/r04_ab_bloom_on_demand.txt
