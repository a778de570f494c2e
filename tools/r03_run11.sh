#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r03_run11.sh v10': the validation of the final tree -- smoke(), the full GPU suite, the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v10}
cd "$R" || exit 1
mkdir -p gpurun_out
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > "gpurun_out/r03_smoke_$tag.txt"; cat "gpurun_out/r03_smoke_$tag.txt"
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl" | tail -6 > "gpurun_out/r03_gpu_tests_$tag.txt"; cat "gpurun_out/r03_gpu_tests_$tag.txt"
timeout 300 python bench.py > "gpurun_out/r03_bench_$tag.json" 2> "gpurun_out/r03_bench_$tag.err"
python - "$tag" <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/r03_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(j["ms_per_step"], j["value"], r["whole_chain"]["frac"], r["kernel"], r["kernel_ms"], r["frac"], r["traffic"], r["traffic_source"], r["valu"]["insts_source"], j["cpu_baseline"]["value"])
PY
