#!/bin/bash
# Which hardware counters separate the fast (~3.33 ms) from the slow (~3.87 ms) mode of the SAME kernel on the SAME box?
# The mode is drawn per process (per allocation), so every counter set is collected in several fresh processes; the
# probe's own event timing (printed by perf_probe.py) tells which mode each process was in.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/${GPU_TAG:-pmc_modes}; mkdir -p $O; export TMPDIR=/tmp
DIMS=${DIMS:-512x512x512}
i=0
while read -r C; do
  i=$((i+1))
  for rep in 1 2 3; do
    ( cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/s${i}_r$rep -o pmc -- \
        python $R/tools/perf_probe.py --dims $DIMS --variants 11 --blocks 576 --reps 12 --modes even,odd,ab > $O/s${i}_r$rep.log 2>&1 )
    echo "set $i rep $rep: $(grep '^dist_a' $O/s${i}_r$rep.log) $(grep '^variant' $O/s${i}_r$rep.log | cut -c30-)"
  done
done <<'SETS'
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_THRASHING_STALL_sum
TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_BUBBLE_sum TCC_REQ_sum
GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
SETS
python - <<PY
import csv, glob, collections, os, re
O='$O'
for d in sorted(glob.glob(O+'/s*_r*/')):
    tag=os.path.basename(d.rstrip('/'))
    agg=collections.defaultdict(lambda:[0.0,0,0.0])
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            kn=row['Kernel_Name']
            if 'fast_' not in kn: continue
            m=re.search(r'fast_(even|row)_kernel<\d+, (\d+)', kn)
            k=(('even' if m.group(1)=='even' else ('ab' if m.group(2)=='0' else 'odd')), row['Counter_Name'])
            agg[k][0]+=float(row['Counter_Value']); agg[k][1]+=1
            try: agg[k][2]+=(int(row['End_Timestamp'])-int(row['Start_Timestamp']))*1e-6
            except Exception: pass
    for (kern,cn),(v,n,t) in sorted(agg.items()):
        print('%-8s %-5s %-46s %.5g   (n=%d, avg dispatch %.3f ms)'%(tag,kern,cn,v/n,n,t/n))
PY
