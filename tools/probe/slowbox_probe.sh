#!/bin/bash
# Is this box one of those where the 256^3 case through the runner path lands in the slow mode (36 instead of 41 GMLUPS)?
# If so, try the placement parameters on it.  (profiles/r02/runner_path_placement.log)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
one() { env "$@" timeout 300 python tools/bench_configs.py --quick --only 1 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['MLUPS_eff'])"; }
base=$(one A=1)
echo "default: $base"
slow=$(python -c "print(1 if float('${base:-0}') < 38500 else 0)")
if [ "$slow" = 1 ] || [ -n "$FORCE_PROBE" ]; then
  echo "second run of the default: $(one A=1)"
  for opt in SLF_PLACEMENT=0 SLF_PLACEMENT_PARTS=8 SLF_PLACEMENT_PARTS=32 SLF_PLACEMENT_PARTS=4 SLF_PLACEMENT_POW2=1 SLF_PLACEMENT_SPAN_GIB=144 SLF_PLACEMENT_SPAN_GIB=24; do
    echo "$opt: $(one $opt)"
  done
  timeout 300 python tools/perf_probe.py --dims 256x256x256 --variants 11 --blocks 576 --modes even,odd,ab 2>&1 | grep variant
  timeout 300 python bench.py --steps 60 --no_cpu_baseline --repeats 1 --prewarm_steps 100 | cut -c1-140
fi
