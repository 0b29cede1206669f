#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
{
for i in 1 2; do
python bench.py --access_pattern AA --repeats 1 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AA zero-filled   ', d['value'], d['roofline']['kernel_ms'])"
SLF_NO_ZEROFILL=1 python bench.py --access_pattern AA --repeats 1 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AA no zero-fill  ', d['value'], d['roofline']['kernel_ms'])"
done
python bench.py --access_pattern AB --repeats 1 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB zero-filled   ', d['value'], d['roofline']['kernel_ms'])"
python tools/perf_probe.py --size 512 --variants 11,11 --blocks 512 --modes even,odd,aa
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/aa_diag.log
