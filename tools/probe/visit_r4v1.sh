export GPU_TAG=r4v1
O=gpurun_out/r4v1; mkdir -p $O
bash tools/gpu.sh host
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/probe/host_time_probe.py 1 2 4 8 2>&1 | grep -v Warning | tee $O/host_time_probe.jsonl
