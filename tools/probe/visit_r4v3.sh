export GPU_TAG=r4v3
O=gpurun_out/r4v3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_slab.py tests/test_gpu_two_ranks.py tests/test_gpu_comm.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_slab.log
for env in "SLF_STEP_PLAN=1" "SLF_STEP_PLAN=0" "SLF_STEP_PLAN=1 SLF_CALC_STREAMS=1"; do
  echo "== $env"
  env $env timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/probe/host_time_probe.py 1 2 4 2>&1 | grep '^{' | tee -a $O/host_time_probe_$(echo $env | tr ' =' '__').jsonl | cut -c1-330
done
