export GPU_TAG=r4v7
O=gpurun_out/r4v7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sc.py -m gpu -x -q -k "indirect" 2>&1 | tail -6 | tee $O/pytest_sc_indirect.log
for v in 11 9 11 9; do echo "SLF_VARIANT=$v"; SLF_VARIANT=$v timeout 600 python tools/bench_configs.py --only 4 2>/dev/null | tee -a $O/configs_variant$v.jsonl | cut -c1-120; done
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
