export GPU_TAG=r4v8
O=gpurun_out/r4v8; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
for i in 1 2; do timeout 600 python tools/bench_configs.py --only 4 2>/dev/null | tee -a $O/configs.jsonl | cut -c1-120; done
