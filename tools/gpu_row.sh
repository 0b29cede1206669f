#!/bin/bash
# General-path (node map) sweep: whole-row kernels (SLF_VARIANT bit 8) vs per-node kernel, same box visit.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { # label variant args...
  L=$1; V=$2; shift 2
  SLF_VARIANT=$V python examples/ldc_3d.py --mode=benchmark --max_iters=500 --benchmark_sample_from=200 "$@" 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/$L variant=$V /"
}
{
for V in 11 3 0; do
  run "ldc3d_512_bgk_AA" $V --lat_nx=512 --lat_ny=512 --lat_nz=512 --access_pattern=AA
  run "ldc3d_512_bgk_AB" $V --lat_nx=512 --lat_ny=512 --lat_nz=512 --access_pattern=AB
done
for V in 11 3; do
  run "ldc3d_512_mrt_AA" $V --lat_nx=512 --lat_ny=512 --lat_nz=512 --access_pattern=AA --model=mrt
  run "ldc3d_384_f64_AA" $V --lat_nx=384 --lat_ny=384 --lat_nz=384 --access_pattern=AA --precision=double
  SLF_VARIANT=$V python examples/ldc_2d.py --mode=benchmark --max_iters=3000 --benchmark_sample_from=1000 --lat_nx=1024 --lat_ny=1024 --access_pattern=AA 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/ldc2d_1024_AA variant=$V /"
  SLF_VARIANT=$V python examples/poiseuille_3d.py --mode=benchmark --max_iters=500 --benchmark_sample_from=200 --lat_nx=256 --lat_ny=256 --lat_nz=512 --access_pattern=AA 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/pipe3d_256x256x512_AA variant=$V /"
done
} 2>&1 | tee gpurun_out/row_general.log
