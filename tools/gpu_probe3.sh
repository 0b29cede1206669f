#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
echo "== with collision"
python tools/perf_probe.py --size 512 --variants 8,9,24,40,9 --blocks 576 --reps 100 --modes even,odd,ab
echo "== no relaxation (pure traffic)"
python tools/perf_probe.py --size 512 --variants 8,9,24,40 --blocks 576 --reps 100 --modes even,odd,ab --norelax
} 2>&1 | tee gpurun_out/probe4.log
