#!/bin/bash
# Same box: the 512^3 cavity through the runner path (quick) and bench.py's 512^3 periodic box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python tools/bench_configs.py --quick --only 2b 2>/dev/null | grep config | cut -c1-110
timeout 300 python bench.py --steps 40 --no_cpu_baseline --repeats 1 --prewarm_steps 60 --no_gpu_state 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['config']['candidates_mlups'], d['config']['placement'])"
