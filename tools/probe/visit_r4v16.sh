export GPU_TAG=r4v16
O=gpurun_out/r4v16; mkdir -p $O
V=sailfish_amd/lib/variants
# the rest of the GPU suite (the run before stopped at a comparison of padding NaNs in a new test)
timeout 900 python -m pytest tests/test_gpu_slab.py tests/test_gpu_two_ranks.py -m gpu -q 2>&1 | tail -6 | tee $O/pytest_rest.log
# fused Shan-Chen sweep: order of the memory phases (0 = as committed, 1 = lattice 0 requested before the stencil,
# 2 = + lattice 1 requested ahead of lattice 0's stores); parity of every variant, then A/B/C twice round
for o in 1 2; do
  SLF_LIBRARY=$PWD/$V/libsailfish_hip_o$o.so timeout 600 python -m pytest tests/test_gpu_sc.py tests/test_gpu_fullsize.py -m gpu -q -k "sc or shan" 2>&1 | tail -3 | tee $O/pytest_sc_o$o.log
done
for rep in 1 2; do
  for o in 0 1 2; do
    SLF_LIBRARY=$PWD/$V/libsailfish_hip_o$o.so timeout 300 python tools/bench_configs.py --only 4 | sed "s/^{/{\"order\": $o, /" | tee -a $O/configs_sc_order.jsonl | cut -c1-120
  done
done
( cd /tmp && SLF_LIBRARY=$GRAFT_REPO_ROOT/$V/libsailfish_hip_o2.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_o2 -o t -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --quick --only 4 > /dev/null 2>&1 )
f=$(find $O/trace_o2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_sc_order2.csv && head -6 $f | cut -c1-200
rm -rf $O/trace_o2
