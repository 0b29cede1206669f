#!/bin/bash
# One GPU-box visit of round 5+: `gpurun -- 'bash tools/visit.sh <name>'` runs the block named <name> below and writes under
# gpurun_out/<name>/.  (Rounds 3-4 kept one script per visit under tools/probe/; this file replaces them: one block per
# visit, newest last.  tools/gpu.sh holds the reusable tasks.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
NAME=${1:-none}; export GPU_TAG=$NAME; O=gpurun_out/$NAME; mkdir -p $O; export TMPDIR=/tmp
case $NAME in
  r5v1)   # new tests of round 5 (3-D physics regressions, config 4 with 8 subdomains, rank tests) + a bench line
    bash tools/gpu.sh host smoke
    ( time timeout 900 python -m pytest tests/test_gpu_physics3d.py -m gpu -q -s --durations=12 ) > $O/pytest_physics3d.log 2>&1; grep -v "^$" $O/pytest_physics3d.log | tail -60
    ( time timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k eight_subdomains --durations=4 ) > $O/pytest_config4_8.log 2>&1; tail -25 $O/pytest_config4_8.log
    ( time timeout 1200 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_face_kernels.py -m gpu -q -x --durations=6 ) > $O/pytest_ranks.log 2>&1; tail -15 $O/pytest_ranks.log
    timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $O/bench_driver_cmd.json; cut -c1-700 $O/bench_driver_cmd.json
    ;;
  *) echo "unknown visit $NAME" ;;
esac
