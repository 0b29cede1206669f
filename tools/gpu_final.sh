#!/bin/bash
# Round-end measurement visit: full GPU test suite, smoke, bench line, kernel traces, PMC traffic of the
# node-map (cavity) path, per-configuration table.  Everything lands in gpurun_out/final/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | grep -m2 -i "card series\|GPU\[0\]" > $O/host.txt; nproc >> $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt
timeout 900 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -12 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.log
timeout 600 python bench.py 2>&1 | tail -1 | tee $O/bench.json
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -o trace -- python $R/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $O/trace_bench.log 2>&1 )
for f in $(find $O/trace_bench -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench.csv; head -6 $f | cut -c1-200; done
CAV="$R/examples/ldc_3d.py --lat_nx=512 --lat_ny=512 --lat_nz=512 --access_pattern=AA --visc=0.0256 --quiet"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cavity -o trace -- python $CAV --max_iters=61 > $O/trace_cavity.log 2>&1 )
for f in $(find $O/trace_cavity -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_cavity.csv; head -5 $f | cut -c1-200; done
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_cavity_p$i -o pmc -- python $CAV --max_iters=13 > $O/pmc_cavity_p$i.log 2>&1 )
done
python - <<'PY' | tee $O/pmc_cavity_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob('gpurun_out/final/pmc_cavity_p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        kn = row['Kernel_Name']
        if 'row_kernel' not in kn and 'even_kernel' not in kn:
            continue
        k = (kn.split('(')[0][-60:], row['Counter_Name'])
        agg[k][0] += float(row['Counter_Value']); agg[k][1] += 1
tot = collections.defaultdict(dict)
for (kn, cn), (v, n) in sorted(agg.items()):
    print('%-62s %-24s %.6g (n=%d)' % (kn, cn, v / n, n)); tot[kn][cn] = v / n
n3 = 512 ** 3
for kn, c in tot.items():
    if 'TCC_EA0_RDREQ_sum' in c and 'WRITE_SIZE' in c:
        rd, wr = c['TCC_EA0_RDREQ_sum'] * 128, c['WRITE_SIZE'] * 1024
        print('%s: HBM read %.4g B + written %.4g B = %.4g B per launch; algorithmic 512^3 x (152 + 4 map) = %.4g B'
              % (kn, rd, wr, rd + wr, n3 * 156.0))
PY
timeout 600 python tools/bench_configs.py --out $O/configs.jsonl 2>&1 | grep -v amdgpu.ids | tail -8
ls $O
