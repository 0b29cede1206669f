# Round-4 evidence visit (third: after the Shan-Chen changes and the XCD row regrouping): what DESIGN.md / profiles/traffic.json quote.
export GPU_TAG=r4final3
O=gpurun_out/r4final3; mkdir -p $O
bash tools/gpu.sh host smoke
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
timeout 900 python bench.py 2>&1 | tail -1 > $O/bench_final.json; cut -c1-600 $O/bench_final.json
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $O/bench_driver_cmd_final.json; cut -c1-300 $O/bench_driver_cmd_final.json
for pat in AA; do
  BENCH_ARGS="--access_pattern $pat" bash tools/gpu.sh trace; cp $O/kernel_stats.csv $O/kernel_stats_${pat}_final.csv; rm -rf $O/trace
done
timeout 1500 python tools/bench_configs.py 2>/dev/null | grep '^{' > $O/configs_final.jsonl; cut -c1-140 $O/configs_final.jsonl
TRACE_CONFIGS="4" bash tools/gpu.sh pmccfg tracecfg; rm -rf $O/pmc_cfg*/ $O/trace_cfg*/
bash tools/gpu.sh torchrun; mv $O/torchrun.jsonl $O/torchrun_final.jsonl
