export GPU_TAG=r4v11
O=gpurun_out/r4v11; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
