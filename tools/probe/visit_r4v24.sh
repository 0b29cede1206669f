export GPU_TAG=r4v24
O=gpurun_out/r4v24; mkdir -p $O
V=$PWD/sailfish_amd/lib/variants
for k in 4 16; do
  SLF_LIBRARY=$V/libsailfish_hip_x$k.so timeout 600 python -m pytest tests/test_gpu_sc.py tests/test_gpu_fullsize.py -m gpu -q -x -k "vs_oracle or equals or shan or multi_subdomain" 2>&1 | tail -2 | tee -a $O/pytest_sc_xcd.log
done
# rows of a block of 8 K handed to the XCDs K consecutive rows each (x4, x16) against every eighth row (as built)
for rep in 1 2; do
  timeout 300 python tools/bench_configs.py --only 4 | sed "s/^{/{\"xcd_rows\": 0, /" | tee -a $O/configs_sc_xcd_rows.jsonl | cut -c1-120
  for k in 4 16; do
    SLF_LIBRARY=$V/libsailfish_hip_x$k.so timeout 300 python tools/bench_configs.py --only 4 | sed "s/^{/{\"xcd_rows\": $k, /" | tee -a $O/configs_sc_xcd_rows.jsonl | cut -c1-120
  done
done
for k in 16; do
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    ( cd /tmp && SLF_LIBRARY=$V/libsailfish_hip_x$k.so timeout 300 rocprofv3 --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_x$k/p$i -o pmc -- env SLF_PLACEMENT_TUNE=0 python $GRAFT_REPO_ROOT/tools/bench_configs.py --quick --only 4 > /dev/null 2>&1 )
  done
  python tools/pmc_summary.py $O/pmc_x$k | grep -A5 "sc_fused" | tee $O/pmc_summary_sc_xcd_rows_$k.txt
  rm -rf $O/pmc_x$k
done
