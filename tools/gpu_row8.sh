#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
# 11: loads issued before the node map (speculative) | 75: loads predicated on the node map
for G in fluid walls pipe; do
python tools/perf_probe.py --size 512 --variants 11,75,11,75 --blocks 512 --modes odd,ab --general $G
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/row_probe8.log
