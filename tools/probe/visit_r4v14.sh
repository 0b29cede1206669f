export GPU_TAG=r4v14
O=gpurun_out/r4v14; mkdir -p $O
for i in 1 2; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 200 --warmup 10 --no_cpu_baseline --force_distributed --scaling strong --domain 128x512x512 --axis x 2>&1 | tail -1 | tee -a $O/torchrun_x_128.jsonl | python -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print(d['value'], d['median_value'], c['candidates_mlups'], c.get('halo_overlap_frac'), c.get('halo_exposed_ms'), c['per_rank'][0], c['validated'])"
done
