#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run6.sh': no contraction + the separate multiplies and adds in the march, but the hardware transcendentals / 1-ulp-corrected division kept
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
B="--steps 60 --warmup 30 --no-cpu-baseline --no-pass-breakdown --no-kernel-sweep --overlap 3"
for n in fast nofma fast nofma; do
    MIFX_LIB_PATH=$R/diligentfx_amd/variants/$n.so timeout 200 python bench.py $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], d['ms_per_step_median'])"
done | tee gpurun_out/r04_ab_nofma_vs_fast.txt
rm -f /tmp/nofma.jsonl
MIFX_LIB_PATH=$R/diligentfx_amd/variants/nofma.so MIFX_PARITY_LOG=/tmp/nofma.jsonl MIFX_PARITY_MEASURE=1 timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -3
cp /tmp/nofma.jsonl gpurun_out/r04_parity_nofma.jsonl
