export GPU_TAG=r4v9
O=gpurun_out/r4v9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
for i in 1 2 3; do
  for lib in "" "$PWD/sailfish_amd/lib/libsailfish_hip_shfl.so"; do
    SLF_LIBRARY=$lib timeout 600 python bench.py --steps 100 --no_cpu_baseline --no_runner_path --no_validate --no_gpu_state 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('lib=%s' % ('shfl' if '$lib' else 'dpp'), d['value'], d['median_value'], c['candidates_mlups'], d['roofline']['kernel_ms'])" | tee -a $O/ab_dpp.txt
  done
done
