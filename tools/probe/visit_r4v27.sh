export GPU_TAG=r4v27
O=gpurun_out/r4v27; mkdir -p $O
# HBM traffic of the x-slab pair's kernels with the rows as they come (0) and regrouped for the XCDs (5)
for s in 0 5; do
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    ( cd /tmp && SLF_XCD_ROWS_LOG2=$s timeout 300 rocprofv3 --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_s$s/p$i -o pmc -- env SLF_PLACEMENT_TUNE=0 python $GRAFT_REPO_ROOT/tools/bench_configs.py --quick --only 3b > /dev/null 2>&1 )
  done
  python tools/pmc_summary.py $O/pmc_s$s > $O/pmc_summary_cfg3b_xcd_rows_$s.txt; grep -A5 "row_kernel" $O/pmc_summary_cfg3b_xcd_rows_$s.txt | head -8
  rm -rf $O/pmc_s$s
done
