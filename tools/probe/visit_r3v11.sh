export GPU_TAG=r3v11
O=gpurun_out/r3v11; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sc.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_kat.py tests/test_gpu_runner.py tests/test_gpu_fullsize.py -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
echo "=== MRT cavity, 7 waves (default)"
for i in 1 2; do timeout 600 python tools/bench_configs.py --only 2 2>/dev/null | tee -a $O/configs_mrt7.jsonl | cut -c1-120; done
echo "=== MRT cavity, 8 waves"
for i in 1 2; do SLF_LIBRARY=$PWD/sailfish_amd/lib/variants/libsailfish_hip_mrt8.so timeout 600 python tools/bench_configs.py --only 2 2>/dev/null | tee -a $O/configs_mrt8.jsonl | cut -c1-120; done
echo "=== MRT periodic box 512"
timeout 600 python bench.py --model mrt --steps 100 --warmup 10 --no_cpu_baseline --no_runner_path 2>/dev/null | tee $O/bench_mrt.json | cut -c1-200
TRACE_CONFIGS="2" bash tools/gpu.sh sqcfg 2>&1 | grep -v "^$" | grep "^void\|per wave\|of SQ_BUSY" | head -20
TRACE_CONFIGS="2" bash tools/gpu.sh tracecfg 2>&1 | tail -8 | cut -c1-200
rm -rf $O/sq_cfg*/p*
