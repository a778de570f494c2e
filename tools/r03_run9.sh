#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r03_run9.sh': the GPU parity suite with the outlier fraction of every comparison logged, shipped build and strict build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
rm -f /tmp/shipped.jsonl /tmp/strict.jsonl
MIFX_PARITY_LOG=/tmp/shipped.jsonl timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
MIFX_LIB_PATH=$R/diligentfx_amd/variants/strict.so MIFX_PARITY_LOG=/tmp/strict.jsonl MIFX_PARITY_MEASURE=1 timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -3
python tools/parity_table.py /tmp/shipped.jsonl /tmp/strict.jsonl > gpurun_out/r03_parity_outliers_strict_vs_shipped.txt
grep -i "end to end\|SSR output" gpurun_out/r03_parity_outliers_strict_vs_shipped.txt
wc -l gpurun_out/r03_parity_outliers_strict_vs_shipped.txt
