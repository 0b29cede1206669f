#!/bin/bash
# Row pitch / array stride experiments for row lengths that run below the 512^3 figure.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { # size align pad
  SLF_MEM_ALIGNMENT=$2 SLF_DIST_PAD=$3 python bench.py --size $1 --repeats 1 --prewarm_steps 60 --steps 60 --warmup 6 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('size $1 align $2 pad $3  %s' % d['config']['candidates_mlups'])"
}
{
run 512 32 0
run 640 32 0
run 640 64 0
run 640 128 0
run 640 32 4128
run 640 32 1056
run 576 32 0
run 704 32 0
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pad.log
