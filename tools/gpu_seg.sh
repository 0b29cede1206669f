#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for N in 512 640 576 768 384; do
  python bench.py --size $N --repeats 1 --prewarm_steps 60 --steps 60 --warmup 6 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('size $N  %s' % d['config']['candidates_mlups'])"
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/seg.log
