O=gpurun_out/r3v5; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
echo "=== bench default"; ( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real; python -c "
import json; d = json.load(open('$O/bench.json')); c = d['config']
print(d['value'], d['median_value'], d['roofline']['frac'], 'validated', c.get('validated'), 'runner_path', c.get('runner_path'))"; tail -2 $O/bench.err
run_x() { tag=$1; shift
  for pat in AA AB; do env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 200 --warmup 20 --no_cpu_baseline --force_distributed --scaling strong --domain 128x512x512 --axis x --access_pattern $pat --no_validate --no_gpu_state 2>&1 | tail -1 | tee -a $O/torchrun_x_$tag.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$tag', c['access_pattern'], d['value'], c['per_rank'], c.get('halo_overlap_frac'), c.get('halo_exposed_ms'))"; done; }
echo "=== strong x"
run_x rccl_k4 SLF_XFACE_CHUNKS=4
run_x rccl_k1 SLF_XFACE_CHUNKS=1
run_x rccl_k3 SLF_XFACE_CHUNKS=3
echo "=== runner path spread, placement by measurement"
for i in 1 2 3; do timeout 300 python tools/bench_configs.py --quick --only 1 2>&1 | grep '^{' | tee -a $O/configs_256_tuned.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('256^3 tuned', d['MLUPS_eff'], d.get('placement_tuning'))"; done
for i in 1 2 3; do SLF_PLACEMENT_TUNE=0 timeout 300 python tools/bench_configs.py --quick --only 1 2>&1 | grep '^{' | tee -a $O/configs_256_untuned.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('256^3 untuned', d['MLUPS_eff'])"; done
for i in 1 2; do timeout 400 python tools/bench_configs.py --quick --only 2b 2>&1 | grep '^{' | tee -a $O/configs_cavity_tuned.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cavity tuned', d['MLUPS_eff'], d.get('placement_tuning'))"; done
echo "=== x-slab pair (K=1 in-process)"; timeout 300 python tools/bench_configs.py --quick --only 3 2>&1 | grep '^{' | tee $O/configs_xslab.jsonl | cut -c1-200
