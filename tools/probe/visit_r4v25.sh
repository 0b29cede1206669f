export GPU_TAG=r4v25
O=gpurun_out/r4v25; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sc.py tests/test_gpu_slab.py tests/test_gpu_fullsize.py tests/test_gpu_runner.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.log
# rows per XCD and block of rows = 1 << SLF_XCD_ROWS_LOG2 (0 = every eighth row, as before this visit)
for rep in 1 2; do
  for s in 0 4 5; do
    for c in 4 3 3b; do
      SLF_XCD_ROWS_LOG2=$s timeout 300 python tools/bench_configs.py --only $c | sed "s/^{/{\"xcd_rows_log2\": $s, /" | tee -a $O/configs_xcd_rows.jsonl | cut -c1-150
    done
  done
done
