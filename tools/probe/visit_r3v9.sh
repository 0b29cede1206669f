export GPU_TAG=r3v9
O=gpurun_out/r3v9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sc.py tests/test_gpu_runner.py tests/test_gpu_slab.py tests/test_gpu_fullsize.py -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
for i in 1 2 3; do
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline > $O/bench_tuned_$i.json 2> $O/bench_tuned_$i.err ) 2>&1 | grep real
python -c "
import json; d = json.load(open('$O/bench_tuned_$i.json')); c = d['config']
print('tuned', d['value'], d['median_value'], c['runs_mlups'], c['candidates_mlups'], c['placement'].get('tuning'), 'validated', c.get('validated'), 'runner_path', c.get('runner_path', {}).get('mlups'))"
done
for i in 1 2; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline --no_placement_tune --no_runner_path --no_validate > $O/bench_untuned_$i.json 2> $O/bench_untuned_$i.err
python -c "
import json; d = json.load(open('$O/bench_untuned_$i.json')); c = d['config']
print('untuned', d['value'], d['median_value'], c['runs_mlups'], c['candidates_mlups'])"
done
