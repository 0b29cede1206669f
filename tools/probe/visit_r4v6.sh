export GPU_TAG=r4v6
O=gpurun_out/r4v6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sc.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_sc.log
for i in 1 2; do timeout 600 python tools/bench_configs.py --only 4 2>/dev/null | tee -a $O/configs.jsonl | cut -c1-200; done
TRACE_CONFIGS=4 bash tools/gpu.sh tracecfg
