export GPU_TAG=r4v15
O=gpurun_out/r4v15; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_runner.py tests/test_gpu_slab.py -m gpu -q 2>&1 | tail -6 | tee $O/pytest.log
