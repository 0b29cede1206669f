export GPU_TAG=r3v12
O=gpurun_out/r3v12; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sc.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_runner.py tests/test_gpu_slab.py tests/test_gpu_fullsize.py -m gpu -q -x --durations=3 > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for c in 2 2b 2c 5a 5b 4; do timeout 600 python tools/bench_configs.py --only $c 2>/dev/null | tee -a $O/configs.jsonl | cut -c1-130; done
