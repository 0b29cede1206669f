#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/perf_probe.py --size 512 --variants 9,13,11,9 --blocks 576 --reps 100 --modes even,ab 2>&1 | tee gpurun_out/probe6.log
python bench.py --steps 200 --warmup 20 2>&1 | tail -2 | tee gpurun_out/bench2.log
