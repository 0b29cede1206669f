O=gpurun_out/r3v4; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
run_x() {  # $1 tag, rest: env assignments
  tag=$1; shift
  for pat in AA AB; do env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 200 --warmup 20 --no_cpu_baseline --force_distributed --scaling strong --domain 128x512x512 --axis x --access_pattern $pat --no_validate --no_gpu_state 2>&1 | tail -1 | tee -a $O/torchrun_x_$tag.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$tag', c['access_pattern'], d['value'], c['per_rank'], c.get('halo_overlap_frac'), c.get('halo_exposed_ms'))"; done; }
echo "=== strong x"
run_x rccl_k4 SLF_XFACE_CHUNKS=4
run_x rccl_k8 SLF_XFACE_CHUNKS=8
run_x rccl_k2 SLF_XFACE_CHUNKS=2
run_x rccl_k1 SLF_XFACE_CHUNKS=1
run_x torch_k4 SLF_XFACE_CHUNKS=4 SLF_HALO_TRANSPORT=torch
echo "=== configs"; timeout 900 python tools/bench_configs.py --quick --only 2,2b,3 2>&1 | grep '^{' | tee $O/configs.jsonl | cut -c1-200
echo "=== with ghost stores"; SLF_X_GHOST_STORES=1 timeout 600 python tools/bench_configs.py --quick --only 2,2b 2>&1 | grep '^{' | tee $O/configs_ghoststores.jsonl | cut -c1-200
echo "=== x-slab pair, 1 chunk"; SLF_XFACE_CHUNKS=1 timeout 300 python tools/bench_configs.py --quick --only 3 2>&1 | grep '^{' | tee $O/configs_xslab_k1.jsonl | cut -c1-200
