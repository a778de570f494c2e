#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run5.sh': the strict build (no contraction anywhere, exact divisions / square roots, separate multiplies and adds in the march) against
# the contracting build: frame time side by side (mode 3), and the outlier fraction of every comparison of the GPU parity suite for both
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
B="--steps 60 --warmup 30 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep --overlap 3"
for n in fast strict fast strict; do
    MIFX_LIB_PATH=$R/diligentfx_amd/variants/$n.so timeout 200 python bench.py $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], d['ms_per_step_median'])"
done | tee gpurun_out/r04_ab_strict_vs_fast.txt
rm -f /tmp/shipped.jsonl /tmp/strict.jsonl
MIFX_LIB_PATH=$R/diligentfx_amd/variants/fast.so MIFX_PARITY_LOG=/tmp/shipped.jsonl MIFX_PARITY_MEASURE=1 timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -3
MIFX_LIB_PATH=$R/diligentfx_amd/variants/strict.so MIFX_PARITY_LOG=/tmp/strict.jsonl MIFX_PARITY_MEASURE=1 timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -3
python tools/parity_table.py /tmp/shipped.jsonl /tmp/strict.jsonl > gpurun_out/r04_parity_outliers_strict_vs_fast.txt
cp /tmp/strict.jsonl gpurun_out/r04_parity_strict.jsonl
wc -l gpurun_out/r04_parity_outliers_strict_vs_fast.txt
