export GPU_TAG=r3final
bash tools/gpu.sh host tests bench
BENCH_ARGS="--access_pattern AB" bash tools/gpu.sh trace pmc
mv gpurun_out/r3final/kernel_stats.csv gpurun_out/r3final/kernel_stats_AB.csv; mv gpurun_out/r3final/pmc_summary.txt gpurun_out/r3final/pmc_summary_AB.txt; rm -rf gpurun_out/r3final/pmc gpurun_out/r3final/trace
BENCH_ARGS="--access_pattern AA" bash tools/gpu.sh trace pmc
mv gpurun_out/r3final/kernel_stats.csv gpurun_out/r3final/kernel_stats_AA.csv; mv gpurun_out/r3final/pmc_summary.txt gpurun_out/r3final/pmc_summary_AA.txt; rm -rf gpurun_out/r3final/pmc gpurun_out/r3final/trace
bash tools/gpu.sh torchrun configs
TRACE_CONFIGS="2b 4" bash tools/gpu.sh pmccfg
rm -rf gpurun_out/r3final/pmc_cfg* gpurun_out/r3final/trace_cfg*
O=gpurun_out/r3final
for i in 1 2 3; do timeout 300 python tools/bench_configs.py --quick --only 1 2>&1 | grep '^{' | tee -a $O/configs_256_tuned.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('256^3 tuned', d['MLUPS_eff'], d.get('placement_tuning'))"; done
