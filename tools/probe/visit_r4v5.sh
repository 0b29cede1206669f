export GPU_TAG=r4v5
O=gpurun_out/r4v5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_face_kernels.py tests/test_gpu_runner.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_face.log
for c in 2 2b 1; do timeout 600 python tools/bench_configs.py --only $c 2>/dev/null | tee -a $O/configs.jsonl | cut -c1-220; done
timeout 900 python bench.py 2>&1 | tail -1 | tee $O/bench.json | cut -c1-900
for st in 2 1; do
  for mode in "--scaling weak" "--scaling strong --domain 1024x512x512 --axis z"; do
    SLF_CALC_STREAMS=$st timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
      bench.py --gpus 1 --steps 100 --warmup 10 --no_cpu_baseline --force_distributed --no_validate $mode 2>&1 | tail -1 | tee -a $O/torchrun_streams$st.jsonl | python -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('streams $st', d['value'], d['median_value'], c['candidates_mlups'], c.get('halo_overlap_frac'), c.get('halo_exposed_ms'), c['per_rank'][0].get('host_ms_median'), c['per_rank'][0].get('sweep_only_ms'))"
  done
done
