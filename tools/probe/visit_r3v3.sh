O=gpurun_out/r3v3; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
echo "=== configs"; timeout 900 python tools/bench_configs.py --quick --only 2,3,4 2>&1 | grep '^{' | tee $O/configs.jsonl | cut -c1-200
echo "=== SC no band"; SLF_SC_BAND=0 timeout 300 python tools/bench_configs.py --quick --only 4 2>&1 | grep '^{' | tee $O/configs_sc_noband.jsonl | cut -c1-200
echo "=== x-slab pair without signals"; SLF_XFACE_SIGNALS=0 timeout 300 python tools/bench_configs.py --quick --only 3 2>&1 | grep '^{' | tee $O/configs_xslab_nosignals.jsonl | cut -c1-200
echo "=== torchrun strong x"; for sig in 1 0; do for pat in AA AB; do SLF_XFACE_SIGNALS=$sig timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --no_cpu_baseline --force_distributed --scaling strong --domain 128x512x512 --axis x --access_pattern $pat --no_validate --no_gpu_state 2>&1 | tail -1 | tee -a $O/torchrun_x_sig$sig.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('signals=$sig', c['access_pattern'], d['value'], c['per_rank'], c.get('halo_overlap_frac'), c.get('halo_exposed_ms'))"; done; done
echo "=== kernel trace cavity BGK + SC"
for c in 2b 4; do ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_cfg$c -o trace -- python $OLDPWD/tools/bench_configs.py --quick --only $c > $OLDPWD/$O/trace_cfg$c.log 2>&1 ); f=$(find $O/trace_cfg$c -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_cfg$c.csv; head -6 $f | cut -c1-220; done
