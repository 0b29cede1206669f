export GPU_TAG=r3final2
O=gpurun_out/r3final2; mkdir -p $O
bash tools/gpu.sh host tests smoke bench
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2>&1 | grep real
BENCH_ARGS="--access_pattern AB" bash tools/gpu.sh trace pmc
mv $O/kernel_stats.csv $O/kernel_stats_AB.csv; mv $O/pmc_summary.txt $O/pmc_summary_AB.txt; rm -rf $O/pmc $O/trace
BENCH_ARGS="--access_pattern AA" bash tools/gpu.sh trace pmc
mv $O/kernel_stats.csv $O/kernel_stats_AA.csv; mv $O/pmc_summary.txt $O/pmc_summary_AA.txt; rm -rf $O/pmc $O/trace
bash tools/gpu.sh torchrun configs
TRACE_CONFIGS="2 4" bash tools/gpu.sh tracecfg
rm -rf $O/pmc_cfg* $O/trace_cfg* $O/sq_cfg*
