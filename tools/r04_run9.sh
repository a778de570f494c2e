#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r04_run9.sh': the adapters on real frames, the PostFX texture helpers
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
timeout 400 python -m pytest tests/test_adapter_example.py tests/test_gpu_tonemap_prep.py -q -m gpu 2>&1 | tail -12
