export GPU_TAG=r3v8
O=gpurun_out/r3v8; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
TRACE_CONFIGS="2b 4" bash tools/gpu.sh pmccfg 2>&1 | grep -v "TCC_\|^$" | grep "kernel\|bytes/launch" | head -60
rm -rf gpurun_out/r3v8/pmc_cfg*
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2>&1 | grep real; python -c "
import json; d = json.load(open('$O/bench_driver_cmd.json')); c = d['config']
print(d['value'], d['median_value'], d['roofline']['frac'], 'validated', c.get('validated'), 'runner_path', c.get('runner_path', {}).get('mlups'), c.get('runner_path', {}).get('vs_value'))"
