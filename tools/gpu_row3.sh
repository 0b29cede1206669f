#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
python tools/perf_probe.py --size 512 --variants 11 --blocks 512 --modes even,odd,ab
for G in fluid walls pipe; do
python tools/perf_probe.py --size 512 --variants 11,75,3,67 --blocks 512 --modes even,odd,ab --general $G
done
python tools/perf_probe.py --size 256 --variants 11,75,3 --blocks 256 --modes even,odd,ab --general pipe
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/row_probe3.log
