export GPU_TAG=r3v13
O=gpurun_out/r3v13; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_runner.py tests/test_gpu_slab.py tests/test_gpu_fullsize.py tests/test_gpu_examples.py -m gpu -q -x --durations=3 > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for rows in 1 0; do
  echo "=== SLF_ROWS_PER_WG=$rows (0 = default: eight-wave workgroups)"
  export SLF_ROWS_PER_WG=$rows
  for i in 1 2; do timeout 600 python tools/bench_configs.py --only 3 2>/dev/null | tee -a $O/configs_rows$rows.jsonl | cut -c1-140; done
  for pat in AA AB; do
    timeout 600 python examples/ldc_3d.py --mode=benchmark --lat_nx=256 --lat_ny=256 --lat_nz=256 --max_iters=2000 --visc=0.0256 --access_pattern=$pat 2>&1 | tail -2 | tee -a $O/ldc256_rows$rows.txt
    timeout 600 python examples/ldc_3d.py --mode=benchmark --lat_nx=128 --lat_ny=512 --lat_nz=512 --max_iters=1500 --visc=0.0256 --access_pattern=$pat 2>&1 | tail -2 | tee -a $O/ldc128_rows$rows.txt
  done
done
