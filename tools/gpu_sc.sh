#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sc.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_sc.log
for P in AA AB; do
python examples/binary_fluid/sc_separation_3d.py --lat_nx=256 --lat_ny=256 --lat_nz=256 --access_pattern=$P --mode=benchmark --max_iters=220 --perf_stats_every=100 --seed=1234 2>&1 | grep -E "speed|MLUPS|rror" | tail -3 | sed "s/^/SC256 $P /"
done | tee gpurun_out/sc_bench.log
