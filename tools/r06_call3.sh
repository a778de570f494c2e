#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r06_v1 TEST_TIMEOUT=1500 bash tools/gpu_run.sh tests
grep -E "FAILED|Error" gpurun_out/gpu_tests_full.txt | head -20
python bench.py --overlap 0 --no-cpu-baseline --no-stage-lines --no-pass-breakdown --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one stream', d['ms_per_step'], d['roofline']['per_kernel_ms'])"
python bench.py --no-cpu-baseline --no-stage-lines --no-pass-breakdown --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes', d['ms_per_step'], d.get('overlap_verified',{}).get('frames_that_differed'))"
