#!/bin/bash
# Counters of the SAME kernel on fast and slow allocations inside one process (tools/probe/alloc_probe.py, 8 arrays).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/${GPU_TAG:-alloc_pmc}; mkdir -p $O; export TMPDIR=/tmp
i=0
while read -r C; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/s$i -o pmc -- \
      python $R/tools/probe/alloc_probe.py --arrays 8 --rounds 1 --reps 8 > $O/s$i.log 2>&1 )
  grep "^round" $O/s$i.log | cut -c1-140
  python - <<PY
import csv, glob, collections
rows=[]
for f in glob.glob('$O/s$i/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'fast_' in r['Kernel_Name']:
            rows.append(r)
# dispatch order -> (array, even/odd): per array 12 even launches then 12 odd launches
ids=sorted(set(int(r['Dispatch_Id']) for r in rows))
pos=dict((d,k) for k,d in enumerate(ids))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=pos[int(r['Dispatch_Id'])]
    arr, ph = k//24, (k%24)//12
    if arr>=8: continue
    agg[(arr,'even' if ph==0 else 'odd')][r['Counter_Name']].append(float(r['Counter_Value']))
    agg[(arr,'even' if ph==0 else 'odd')]['ms'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-6)
for key in sorted(agg):
    d=agg[key]
    print('array %d %-4s'%key, ' '.join('%s=%.4g'%(c.replace('_sum',''), sum(v)/len(v)) for c,v in sorted(d.items())))
PY
done <<'SETS'
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
SETS
