#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tuned" 2>&1 | tail -15 | tee gpurun_out/pytest_tuned.log
timeout 600 python tools/perf_probe.py --size 512 --variants 0,1,9 --blocks 576 --reps 200 2>&1 | tee gpurun_out/probe2.log
