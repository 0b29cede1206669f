export GPU_TAG=r3final3
O=gpurun_out/r3final3; mkdir -p $O
( time timeout 1100 python -m pytest tests -m gpu -q --durations=12 ) > $O/pytest_gpu.log 2>&1; tail -18 $O/pytest_gpu.log
bash tools/gpu.sh smoke bench
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --no_cpu_baseline --force_distributed --scaling strong --domain 1024x512x512 --axis x 2>&1 | tail -1 | tee $O/torchrun_x.json | python -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('strong x', d['value'], c['candidates_mlups'], c.get('halo_overlap_frac'), c.get('halo_exposed_ms'), c.get('validated'))"
for c in 3 3b 1; do timeout 600 python tools/bench_configs.py --only $c 2>/dev/null | tee -a $O/configs.jsonl | cut -c1-150; done
