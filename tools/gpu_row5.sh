#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -x -q 2>&1 | tail -2
python tools/perf_probe.py --size 512 --variants 11,267,3 --blocks 512 --modes even,odd,ab
for G in fluid walls pipe; do
python tools/perf_probe.py --size 512 --variants 11,75,3,11,75 --blocks 512 --modes even,odd,ab --general $G
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/row_probe5.log
