#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tuned or periodic_box_f32 or region or large" 2>&1 | tail -5 | tee gpurun_out/pytest_tuned2.log
python tools/perf_probe.py --size 512 --variants 9,11,13 --blocks 512 --reps 100 --modes even,odd,ab 2>&1 | tee gpurun_out/probe7.log
python tools/perf_probe.py --size 512 --variants 11 --blocks 512 --reps 100 --modes even,odd,ab --noalign 2>&1 | tee -a gpurun_out/probe7.log
python bench.py --steps 200 --warmup 20 --no_cpu_baseline 2>&1 | tail -1 | tee gpurun_out/bench3.log
