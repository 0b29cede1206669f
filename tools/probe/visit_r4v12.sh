export GPU_TAG=r4v12
O=gpurun_out/r4v12; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_slab.py -m gpu -q 2>&1 | tail -12 | tee $O/pytest_ranks.log
for i in 1 2; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/probe/host_time_probe.py 2 4 2>&1 | grep '^{' | tee -a $O/host_time_probe_final.jsonl | cut -c1-200
done
