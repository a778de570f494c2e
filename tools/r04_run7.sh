#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run7.sh': the whole GPU suite on the round-4 default build (no contraction, exact march, tightened budgets) + smoke + the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r04_gpu_tests_v13.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 300 python bench.py --overlap 3 2>/dev/null | tail -1 > gpurun_out/r04_bench_v13_ov3.json; python -c "import json; d=json.load(open('gpurun_out/r04_bench_v13_ov3.json')); print('ov3', d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['chain_hbm_frac'])"
timeout 300 python bench.py --overlap 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_v13_ov2.json; python -c "import json; d=json.load(open('gpurun_out/r04_bench_v13_ov2.json')); print('ov2', d['ms_per_step'], d['value'])"
