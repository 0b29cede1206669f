export GPU_TAG=r4v20
O=gpurun_out/r4v20; mkdir -p $O
V=$PWD/sailfish_amd/lib/variants
SLF_LIBRARY=$V/libsailfish_hip_nt.so timeout 600 python -m pytest tests/test_gpu_sc.py -m gpu -q -x -k "vs_oracle or equals" 2>&1 | tail -3 | tee $O/pytest_sc_nt.log
# a = as built (plain density stores), nt = densities stored non-temporally; mode 1 = round-3 kernels for reference
for rep in 1 2; do
  SLF_SC_FUSED=1 timeout 300 python tools/bench_configs.py --only 4 | sed "s/^{/{\"lib\": \"a\", \"SLF_SC_FUSED\": 1, /" | tee -a $O/configs_sc_nt.jsonl | cut -c1-140
  timeout 300 python tools/bench_configs.py --only 4 | sed "s/^{/{\"lib\": \"a\", \"SLF_SC_FUSED\": 2, /" | tee -a $O/configs_sc_nt.jsonl | cut -c1-140
  SLF_LIBRARY=$V/libsailfish_hip_nt.so timeout 300 python tools/bench_configs.py --only 4 | sed "s/^{/{\"lib\": \"nt\", \"SLF_SC_FUSED\": 2, /" | tee -a $O/configs_sc_nt.jsonl | cut -c1-140
done
