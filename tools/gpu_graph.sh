#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_runner.py tests/test_gpu_sc.py -x -q -k "not erturk and not reference_record" 2>&1 | tail -3
for G in "" "--nohip_graphs"; do
  python examples/ldc_2d.py --mode=benchmark --max_iters=20000 --benchmark_sample_from=4000 --lat_nx=256 --lat_ny=256 --access_pattern=AA $G 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/ldc2d_256 AA $G /"
  python examples/ldc_2d.py --mode=benchmark --max_iters=20000 --benchmark_sample_from=4000 --lat_nx=256 --lat_ny=256 --access_pattern=AB $G 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/ldc2d_256 AB $G /"
  python examples/ldc_2d.py --mode=benchmark --max_iters=8000 --benchmark_sample_from=2000 --lat_nx=1024 --lat_ny=1024 --access_pattern=AA $G 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/ldc2d_1024 AA $G /"
  python examples/ldc_3d.py --mode=benchmark --max_iters=4000 --benchmark_sample_from=1000 --lat_nx=64 --lat_ny=64 --lat_nz=64 --access_pattern=AA $G 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/ldc3d_64 AA $G /"
  python examples/ldc_3d.py --mode=benchmark --max_iters=1500 --benchmark_sample_from=500 --lat_nx=256 --lat_ny=256 --lat_nz=256 --access_pattern=AA $G 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/ldc3d_256 AA $G /"
  python examples/sc_phase_separation.py --mode=benchmark --max_iters=8000 --benchmark_sample_from=2000 $G 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/sc_single_2d $G /"
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/graphs.log
