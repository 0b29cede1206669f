#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_sc.py -x -q -k "not reference_record and not phase_separation_and_mass" 2>&1 | tail -3
for V in 11 3 11 3; do
  for P in AA AB; do
  SLF_VARIANT=$V python examples/binary_fluid/sc_separation_3d.py --mode=benchmark --max_iters=600 --benchmark_sample_from=200 --lat_nx=256 --lat_ny=256 --lat_nz=256 --access_pattern=$P 2>&1 | grep -E "Total MLUPS|rror" | sed "s/^/sc_binary_256 $P variant=$V /"
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sc_row.log
