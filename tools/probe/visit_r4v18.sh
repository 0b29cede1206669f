export GPU_TAG=r4v18
O=gpurun_out/r4v18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sc.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sc or shan or densities" 2>&1 | tail -8 | tee $O/pytest_sc.log
# binary Shan-Chen 256^3: 1 = fused sweep reading the stored velocity (rounds 2-4), 2 = densities pass + sweep with its own moments
for rep in 1 2; do
  for m in 1 2; do
    SLF_SC_FUSED=$m timeout 300 python tools/bench_configs.py --only 4 | sed "s/^{/{\"SLF_SC_FUSED\": $m, /" | tee -a $O/configs_sc_local_velocity.jsonl | cut -c1-130
  done
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --quick --only 4 > /dev/null 2>&1 )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_shan_chen_local_velocity.csv && head -6 $f | cut -c1-220
rm -rf $O/trace
