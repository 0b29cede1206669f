#!/bin/bash
# One GPU-box visit: smoke, parity tests, bench, kernel-trace profile.  Outputs under gpurun_out/.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8
nproc; grep -m1 "model name" /proc/cpuinfo
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -25
timeout 900 python bench.py --steps 200 --warmup 20 2>&1 | tee gpurun_out/bench.log | tail -5
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_trace" -o trace -- python "$OLDPWD/bench.py" --steps 100 --warmup 10 --no_cpu_baseline > "$OLDPWD/gpurun_out/bench_prof.log" 2>&1 )
find gpurun_out/prof_trace -name "*stats*" | head; 
for f in $(find gpurun_out/prof_trace -name "*kernel_stats.csv"); do head -12 $f; done
