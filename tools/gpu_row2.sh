#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { # label variant script args...
  L=$1; V=$2; S=$3; shift 3
  SLF_VARIANT=$V python examples/$S --mode=benchmark --max_iters=500 --benchmark_sample_from=200 "$@" 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/$L variant=$V /"
}
{


for V in 11 3; do
  run ldc3d_512_AA $V ldc_3d.py --lat_nx=512 --lat_ny=512 --lat_nz=512 --access_pattern=AA
  run pipe3d_256x256x512_AA $V poiseuille_3d.py --lat_nx=256 --lat_ny=256 --lat_nz=512 --access_pattern=AA
  run ldc3d_1024x256x256_AA $V ldc_3d.py --lat_nx=1024 --lat_ny=256 --lat_nz=256 --access_pattern=AA
  run ldc3d_256_AA $V ldc_3d.py --lat_nx=256 --lat_ny=256 --lat_nz=256 --access_pattern=AA
  run ldc3d_128x512x512_AA $V ldc_3d.py --lat_nx=128 --lat_ny=512 --lat_nz=512 --access_pattern=AA
  SLF_VARIANT=$V python examples/ldc_2d.py --mode=benchmark --max_iters=3000 --benchmark_sample_from=1000 --lat_nx=1024 --lat_ny=1024 --access_pattern=AA 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/ldc2d_1024_AA variant=$V /"
  SLF_VARIANT=$V python examples/ldc_2d.py --mode=benchmark --max_iters=3000 --benchmark_sample_from=1000 --lat_nx=512 --lat_ny=4096 --access_pattern=AA 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/ldc2d_512x4096_AA variant=$V /"
done
} 2>&1 | tee gpurun_out/row_general2.log
