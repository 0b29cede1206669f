#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for M in bgk mrt; do for P in AA AB; do
python examples/ldc_3d.py --lat_nx=512 --lat_ny=512 --lat_nz=512 --visc=0.0637 --model=$M --access_pattern=$P --mode=benchmark --max_iters=220 --perf_stats_every=100 2>&1 | grep -E "speed|MLUPS|Error|error" | tail -3 | sed "s/^/$M $P /"
done; done 2>&1 | tee gpurun_out/general.log
python bench.py --steps 100 --warmup 10 --no_cpu_baseline --model mrt 2>&1 | tail -1 | cut -c1-400 | tee -a gpurun_out/general.log
python bench.py --steps 100 --warmup 10 --no_cpu_baseline --precision double --size 384 2>&1 | tail -1 | cut -c1-400 | tee -a gpurun_out/general.log
