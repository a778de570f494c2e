#!/bin/bash
# round 6, GPU call 1: the new parity points (measure mode), the self-launching bench, the role-interleaved A3 + copy grid
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out
quiet() { grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
(free -g; nproc; rocm-smi --showmeminfo vram 2>/dev/null | head -8) > gpurun_out/r06_box.txt 2>&1
avail=$(free -g | awk '/Mem:/{print $7}')
K="full_size_parity or 3840x2160"
[ "$avail" -gt 80 ] && K="$K or 8k_frame_pair"
MIFX_PARITY_MEASURE=1 MIFX_PARITY_LOG=/tmp/par.jsonl timeout 900 python -m pytest tests/test_gpu_chain.py -k "$K" -x -q -s 2>&1 | quiet | grep -E "frame|passed|failed|rror" > gpurun_out/r06_full_size_parity.txt
tail -12 gpurun_out/r06_full_size_parity.txt
timeout 300 python bench.py > gpurun_out/r06_bench_v1.json 2> gpurun_out/r06_bench_v1.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_v1.json").read().strip().splitlines()[-1])
print("N=1", d["ms_per_step"], d["value"], d.get("overlap_verified"), d["roofline"]["frac"], d["roofline"]["per_kernel_ms"])
PY
timeout 400 python bench.py --gpus 2 --single-gpu --backend gloo --comm torch --width 1920 --height 1080 --steps 6 --warmup 8 > gpurun_out/r06_bench_selflaunch_gloo2.json 2> /tmp/b2.err; echo "gloo2 rc $?"; tail -c 400 /tmp/b2.err | quiet
g++ -shared -fPIC -O1 -std=c++17 -w -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include tests/fake_rccl/fake_rccl.cpp -o /tmp/librccl_fake.so -L /opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib
MIFX_RCCL_PATH=/tmp/librccl_fake.so timeout 400 python bench.py --gpus 2 --single-gpu --width 1920 --height 1080 --steps 6 --warmup 8 > gpurun_out/r06_bench_selflaunch_2procs_standin.json 2> /tmp/b3.err; echo "standin rc $?"; tail -c 400 /tmp/b3.err | quiet
python - <<'PY'
import json
for n in ("r06_bench_selflaunch_gloo2", "r06_bench_selflaunch_2procs_standin"):
    try:
        d = json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
        print(n, d.get("n_gpus"), d.get("ms_per_step"), d.get("shard_verified"), d.get("comm"), d.get("single_gpu_same_frame_ms"), d.get("speedup_vs_single_gpu_same_frame"), d.get("error"))
    except Exception as e:
        print(n, "FAILED", e)
PY
timeout 400 python tools/exp_a3_copy_role.py 2>&1 | quiet | tee gpurun_out/r06_exp_a3_copy_role.txt
