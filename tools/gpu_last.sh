#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/last; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python bench.py 2>&1 | tail -1 | tee $O/bench.json | cut -c1-300
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $R/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $O/trace.log 2>&1 )
for f in $(find $O/trace -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; head -5 $f | cut -c1-160; done
tail -1 $O/trace.log | cut -c1-200
