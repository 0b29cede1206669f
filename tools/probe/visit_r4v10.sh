export GPU_TAG=r4v10
O=gpurun_out/r4v10; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
for c in 3 3b; do
  for plan in 1 0; do echo "config $c SLF_STEP_PLAN=$plan"; SLF_STEP_PLAN=$plan timeout 600 python tools/bench_configs.py --only $c 2>/dev/null | tee -a $O/configs_plan$plan.jsonl | cut -c1-230; done
done
