#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run12.sh': A3 with its taps served from an LDS window (MIFX_A3_WINDOW=1) against the gather: bit-identity, parity suite, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
MIFX_A3_WINDOW=0 timeout 200 python tools/variant_hash.py 2>&1 | grep -v amdgpu.ids > gpurun_out/hash_gather.txt
MIFX_A3_WINDOW=1 timeout 200 python tools/variant_hash.py 2>&1 | grep -v amdgpu.ids > gpurun_out/hash_window.txt
MIFX_A3_WINDOW=1 timeout 200 python tools/variant_hash.py --width 1000 --height 563 2>&1 | grep -v amdgpu.ids > gpurun_out/hash_window_odd.txt
MIFX_A3_WINDOW=0 timeout 200 python tools/variant_hash.py --width 1000 --height 563 2>&1 | grep -v amdgpu.ids > gpurun_out/hash_gather_odd.txt
diff gpurun_out/hash_gather.txt gpurun_out/hash_window.txt && echo "A3 window: bit-identical (1920x1080)"; cat gpurun_out/hash_window.txt | tail -2
diff gpurun_out/hash_gather_odd.txt gpurun_out/hash_window_odd.txt && echo "A3 window: bit-identical (1000x563)"
MIFX_A3_WINDOW=1 timeout 300 python -m pytest tests/test_gpu_ssao.py tests/test_gpu_sharded.py -q -x 2>&1 | tail -3
bash tools/ab_env.sh gather:MIFX_A3_WINDOW=0 window:MIFX_A3_WINDOW=1 gather2:MIFX_A3_WINDOW=0 window2:MIFX_A3_WINDOW=1 | head -12
for w in 0 1; do MIFX_A3_WINDOW=$w timeout 200 python bench.py --steps 60 --warmup 30 --no-cpu-baseline --no-stage-lines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window=$w', d['ms_per_step'], d['value'])"; done
