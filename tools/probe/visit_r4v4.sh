export GPU_TAG=r4v4
O=gpurun_out/r4v4; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 ) > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
