O=gpurun_out/r3v6; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
echo "=== bench default"; ( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real; python -c "
import json; d = json.load(open('$O/bench.json')); c = d['config']
print(d['value'], d['median_value'], d['roofline']['frac'], 'validated', c.get('validated'), 'runner_path', c.get('runner_path'))"; tail -2 $O/bench.err
echo "=== runner path spread, placement by measurement"
for i in 1 2 3 4; do timeout 300 python tools/bench_configs.py --quick --only 1 2>&1 | grep '^{' | tee -a $O/configs_256_tuned.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('256^3 tuned', d['MLUPS_eff'], d.get('placement_tuning'))"; done
for i in 1 2 3; do SLF_PLACEMENT_TUNE=0 timeout 300 python tools/bench_configs.py --quick --only 1 2>&1 | grep '^{' | tee -a $O/configs_256_untuned.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('256^3 untuned', d['MLUPS_eff'])"; done
for i in 1 2; do timeout 400 python tools/bench_configs.py --quick --only 2b 2>&1 | grep '^{' | tee -a $O/configs_cavity_tuned.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cavity tuned', d['MLUPS_eff'], d.get('placement_tuning'))"; done
timeout 400 python tools/bench_configs.py --quick --only 2,4 2>&1 | grep '^{' | tee -a $O/configs_mrt_sc.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['config'][:40], d['MLUPS_eff'], d.get('placement_tuning'))"
