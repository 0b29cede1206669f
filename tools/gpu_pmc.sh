#!/bin/bash
# rocprofv3 runs: kernel trace (csv) + separate PMC passes.  Run from the repo root on the GPU box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
ARGS="--steps 20 --warmup 4 --no_cpu_baseline $BENCH_EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc/trace -o trace -- python $R/bench.py --steps 100 --warmup 10 --no_cpu_baseline $BENCH_EXTRA > $R/gpurun_out/pmc/trace.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc/p$i -o pmc -- python $R/bench.py $ARGS > $R/gpurun_out/pmc/p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob('gpurun_out/pmc/trace/**/*kernel_stats.csv', recursive=True)):
    print('==', f); print(open(f).read()[:1500])
for d in sorted(glob.glob('gpurun_out/pmc/p*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row['Kernel_Name'][:60], row['Counter_Name'])
            agg[k][0] += float(row['Counter_Value']); agg[k][1] += 1
        print('==', f)
        for (kn, cn), (v, n) in sorted(agg.items()):
            if 'sweep' in kn or 'fast' in kn:
                print('  %-60s %-28s avg/launch %.4g  (n=%d)' % (kn, cn, v / n, n))
PY
