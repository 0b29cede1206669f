export GPU_TAG=r4v13
O=gpurun_out/r4v13; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_slab.py tests/test_gpu_two_ranks.py tests/test_gpu_runner.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -6 | tee $O/pytest.log
for b in two all two all; do
echo "SLF_XFACE_BATCHES=$b"
SLF_XFACE_BATCHES=$b timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/probe/host_time_probe.py 4 8 2>&1 | grep '^{' | tee -a $O/host_time_probe_batches_$b.jsonl | cut -c1-330
done
