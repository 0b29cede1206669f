#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for N in 128 256 384 512 640 768; do
  python bench.py --size $N --repeats 1 --prewarm_steps 100 --steps 100 --warmup 10 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bgk f32 %4d^3  %8.1f MLUPS  %6.1f GB/s  %s  %s' % ($N, d['value'], d['roofline']['achieved'], d['config']['access_pattern'], d['config']['candidates_mlups']))"
done
python bench.py --size 512 --model mrt --repeats 1 --prewarm_steps 100 --steps 100 --warmup 10 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mrt f32  512^3  %8.1f MLUPS  %6.1f GB/s  %s' % (d['value'], d['roofline']['achieved'], d['config']['candidates_mlups']))"
python bench.py --size 448 --precision double --repeats 1 --prewarm_steps 100 --steps 100 --warmup 10 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bgk f64  448^3  %8.1f MLUPS  %6.1f GB/s  %s' % (d['value'], d['roofline']['achieved'], d['config']['candidates_mlups']))"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sizes.log
