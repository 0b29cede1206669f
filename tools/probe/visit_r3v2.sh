O=gpurun_out/r3v2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
echo "=== configs default lib"; timeout 900 python tools/bench_configs.py --quick --only 2,2b,2c,5a,3,4 2>&1 | grep '^{' | tee $O/configs_default.jsonl | cut -c1-330
for c in 2 2b 4; do echo "=== config $c lib B (MRT L0 6 waves, SC fused 5 waves)"; SLF_LIBRARY=$PWD/sailfish_amd/lib/alt/libsailfish_hip_b.so timeout 300 python tools/bench_configs.py --quick --only $c 2>&1 | grep '^{' | tee -a $O/configs_libB.jsonl | cut -c1-200; done
echo "=== no row classes"; for c in 2 2b; do SLF_ROW_CLASSES=0 timeout 300 python tools/bench_configs.py --quick --only $c 2>&1 | grep '^{' | tee -a $O/configs_norowclasses.jsonl | cut -c1-200; done
echo "=== SC unfused"; SLF_SC_FUSED=0 timeout 300 python tools/bench_configs.py --quick --only 4 2>&1 | grep '^{' | tee -a $O/configs_sc_unfused.jsonl | cut -c1-200
echo "=== torchrun strong x"; for pat in AA AB; do timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --no_cpu_baseline --force_distributed --scaling strong --domain 128x512x512 --axis x --access_pattern $pat --no_validate 2>&1 | tail -1 | tee -a $O/torchrun_x.jsonl | cut -c1-900; done
