#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for G in fluid walls pipe; do
python tools/perf_probe.py --size 512 --variants 11,75,11,75 --blocks 512 --modes odd --general $G
done
python tools/perf_probe.py --size 512 --variants 267,331,267,331 --blocks 512 --modes odd
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/row_probe6.log
