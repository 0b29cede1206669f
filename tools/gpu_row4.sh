#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
# fluid-only: tuned fast path (11), generic row (267), generic row + direct pull (395), fast scalar NT (3); repeated to see run-to-run spread
python tools/perf_probe.py --size 512 --variants 11,267,395,3,11,267,395 --blocks 512 --modes even,odd,ab
for G in fluid walls pipe; do
python tools/perf_probe.py --size 512 --variants 11,139,3,11,139 --blocks 512 --modes odd --general $G
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/row_probe4.log
