#!/bin/bash
# PMC + trace for AB and AA with the default variant; summaries to gpurun_out/pmc2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
for PAT in AB AA; do
  D=$R/gpurun_out/pmc2/$PAT; mkdir -p $D
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o trace -- python $R/bench.py --steps 100 --warmup 10 --no_cpu_baseline --access_pattern $PAT > $D/trace.log 2>&1
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $C --output-format csv -d $D/p$i -o pmc -- python $R/bench.py --steps 20 --warmup 4 --no_cpu_baseline --access_pattern $PAT > $D/p$i.log 2>&1
  done
  cd $R
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for pat in ('AB', 'AA'):
    print('=====', pat)
    for f in glob.glob('gpurun_out/pmc2/%s/trace/**/*kernel_stats.csv' % pat, recursive=True):
        print(open(f).read()[:900])
    tot = collections.defaultdict(dict)
    for f in sorted(glob.glob('gpurun_out/pmc2/%s/p*/**/*counter_collection.csv' % pat, recursive=True)):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if 'fast' not in row['Kernel_Name'] and 'sweep' not in row['Kernel_Name']:
                continue
            k = (row['Kernel_Name'].split('(')[0][-40:], row['Counter_Name'])
            agg[k][0] += float(row['Counter_Value']); agg[k][1] += 1
        for (kn, cn), (v, n) in sorted(agg.items()):
            print('  %-42s %-26s %.5g (n=%d)' % (kn, cn, v / n, n))
            tot[kn][cn] = v / n
    # HBM bytes per launch: reads = TCC_EA0_RDREQ * 128 B (no 32 B requests seen; equals FETCH_SIZE*1024*2, the
    # gfx950 correction of MI355X_MICROARCH.md), writes = WRITE_SIZE * 1024
    per_kernel = {}
    for kn, c in tot.items():
        if 'TCC_EA0_RDREQ_sum' in c and 'WRITE_SIZE' in c:
            per_kernel[kn] = c['TCC_EA0_RDREQ_sum'] * 128 + c['WRITE_SIZE'] * 1024
    if per_kernel:
        out[pat] = sum(per_kernel.values()) / len(per_kernel)
        print('  traffic per launch (avg over sweep kernels): %.4g B' % out[pat], per_kernel)
json.dump(out, open('gpurun_out/pmc2/traffic_summary.json', 'w'))
PY
