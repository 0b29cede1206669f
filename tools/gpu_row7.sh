#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
# 11 baseline | 75 no map read | 139 fluid-only node_update | 523 fluid-only push flags | 715 all three
python tools/perf_probe.py --size 512 --variants 11,75,139,523,715,11 --blocks 512 --modes odd --general fluid
python tools/perf_probe.py --size 512 --variants 267 --blocks 512 --modes odd
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/row_probe7.log
