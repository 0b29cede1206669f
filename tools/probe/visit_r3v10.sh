export GPU_TAG=r3v10
O=gpurun_out/r3v10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sc.py tests/test_gpu_runner.py -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
TRACE_CONFIGS="2b 2 4" bash tools/gpu.sh sqcfg 2>&1 | grep -v "^$" | grep -A12 "row_kernel\|even_kernel\|sc_fused\|sc_macro" | head -150
rm -rf $O/sq_cfg*/p*
