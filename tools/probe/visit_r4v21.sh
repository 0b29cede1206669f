export GPU_TAG=r4v21
O=gpurun_out/r4v21; mkdir -p $O
V=$PWD/sailfish_amd/lib/variants
timeout 900 python -m pytest tests/test_gpu_sc.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sc or shan or densities" 2>&1 | tail -6 | tee $O/pytest_sc.log
# pull = odd-step populations through aligned loads + lane shifts (as built); nopull = the x-shifted loads
for rep in 1 2; do
  SLF_LIBRARY=$V/libsailfish_hip_nopull.so timeout 300 python tools/bench_configs.py --only 4 | sed "s/^{/{\"lib\": \"nopull\", /" | tee -a $O/configs_sc_row_pull.jsonl | cut -c1-130
  timeout 300 python tools/bench_configs.py --only 4 | sed "s/^{/{\"lib\": \"pull\", /" | tee -a $O/configs_sc_row_pull.jsonl | cut -c1-130
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --quick --only 4 > /dev/null 2>&1 )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_shan_chen_row_pull.csv && head -6 $f | cut -c1-220
rm -rf $O/trace
