O=gpurun_out/r3v14; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_comm.py tests/test_gpu_kat.py tests/test_gpu_examples.py -m gpu -q --durations=25 ) > $O/pytest_subset.log 2>&1; tail -40 $O/pytest_subset.log
