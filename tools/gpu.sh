#!/bin/bash
# One GPU-box visit, parametrised: `gpurun -- 'bash tools/gpu.sh <task> [<task> ...]'`.
# Everything is written under gpurun_out/<tag>/ (tag = $GPU_TAG or "visit"); copy what is to be judged
# into profiles/rNN/ afterwards.  Tasks:
#   host            CPU model / core count / rocm-smi product line
#   smoke           __graft_entry__.smoke()
#   tests [expr]    pytest -m gpu (optionally -k expr via $TEST_K)
#   probe           tools/probe/box_probe.py (clocks + per-launch series; $PROBE_ARGS)
#   bench           python bench.py $BENCH_ARGS
#   trace           rocprofv3 --kernel-trace --stats of bench.py --steps 100 --warmup 10 ($BENCH_ARGS)
#   pmc             separate rocprofv3 --pmc passes of bench.py (HBM traffic; $BENCH_ARGS)
#   torchrun        the N>1 code paths on one GPU: torch.distributed.run --nproc-per-node 1, weak + strong
#   configs         tools/bench_configs.py (all single-GPU BASELINE configs through the host stack)
#   sqcfg           SQ instruction / issue counters of bench_configs.py --quick --only <id>, for id in $TRACE_CONFIGS
#   pmccfg          FETCH_SIZE / WRITE_SIZE passes of bench_configs.py --quick --only <id>, for id in $TRACE_CONFIGS
#   tracecfg        rocprofv3 --kernel-trace --stats of bench_configs.py --quick --only <id>, for id in $TRACE_CONFIGS
#   py <file> ...   python <file> ... (rest of the line)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; TAG=${GPU_TAG:-visit}; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
summarise_trace() {
  for f in $(find $1 -name "*kernel_stats.csv"); do cp $f $2; head -8 $f | cut -c1-200; done
}
while [ $# -gt 0 ]; do
  task=$1; shift
  echo "=== $task"
  case $task in
    host)
      { nproc; grep -m1 "model name" /proc/cpuinfo; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "affinity: $(python -c 'import os;print(len(os.sched_getaffinity(0)))')"; lscpu | grep -i "numa\|socket\|thread" | head -8; rocm-smi --showproductname 2>/dev/null | head -12; rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i partition; amd-smi metric -g 0 2>/dev/null | grep -A2 -i "GFX_0\|MEM_0\|SOCKET_POWER\|THROTTLE" | head -16; } | tee $O/host.txt ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $O/smoke.log | tail -3 ;;
    tests)
      if [ -n "$TEST_K" ]; then timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q -k "$TEST_K" 2>&1 | tee $O/pytest_gpu.log | tail -15
      else timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q 2>&1 | tee $O/pytest_gpu.log | tail -15; fi ;;
    probe)
      timeout 900 python tools/probe/box_probe.py --out $O/box_probe $PROBE_ARGS 2>&1 | tee $O/box_probe.log | tail -60 ;;
    bench)
      timeout 900 python bench.py $BENCH_ARGS 2>&1 | tail -1 | tee $O/bench.json | cut -c1-1500 ;;
    trace)
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- \
          python $R/bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_runner_path --no_validate $BENCH_ARGS > $O/trace.log 2>&1 )
      summarise_trace $O/trace $O/kernel_stats.csv; tail -1 $O/trace.log | cut -c1-400 ;;
    pmc)
      i=0
      # PMC_SIZES_ONLY=1: the two passes profiles/traffic.json is made from, without the request-count cross-check
      for C in "FETCH_SIZE" "WRITE_SIZE" ${PMC_SIZES_ONLY:+SKIP SKIP} "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
        [ "$C" = SKIP ] && break
        i=$((i+1))
        ( cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/pmc/p$i -o pmc -- \
            python $R/bench.py --steps 20 --warmup 4 --no_cpu_baseline --no_runner_path --no_validate --repeats 1 --prewarm_steps 0 $BENCH_ARGS > $O/pmc_p$i.log 2>&1 )
      done
      python tools/pmc_summary.py $O/pmc | tee $O/pmc_summary.txt ;;
    torchrun)
      for mode in "--scaling weak" "--scaling strong --domain 1024x512x512 --axis z" "--scaling strong --domain 1024x512x512 --axis x"; do
        timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
          bench.py --gpus 1 --steps 100 --warmup 10 --no_cpu_baseline --force_distributed $mode 2>&1 | tail -1 | tee -a $O/torchrun.jsonl | cut -c1-1200
      done ;;
    configs)
      timeout 1500 python tools/bench_configs.py $CONFIG_ARGS 2>&1 | tee $O/configs.jsonl | tail -12 ;;
    tracecfg)
      for c in ${TRACE_CONFIGS:-4}; do
        ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg$c -o trace -- \
            env SLF_PLACEMENT_TUNE=0 python $R/tools/bench_configs.py --quick --only $c > $O/trace_cfg$c.log 2>&1 )
        summarise_trace $O/trace_cfg$c $O/kernel_stats_cfg$c.csv; tail -1 $O/trace_cfg$c.log | cut -c1-300
      done ;;
    pmccfg)
      for c in ${TRACE_CONFIGS:-4}; do
        i=0
        for C in "FETCH_SIZE" "WRITE_SIZE"; do
          i=$((i+1))
          ( cd /tmp && timeout 900 rocprofv3 --pmc $C --output-format csv -d $O/pmc_cfg$c/p$i -o pmc -- \
              env SLF_PLACEMENT_TUNE=0 python $R/tools/bench_configs.py --quick --only $c > $O/pmc_cfg${c}_p$i.log 2>&1 )
        done
        python tools/pmc_summary.py $O/pmc_cfg$c | tee $O/pmc_summary_cfg$c.txt
      done ;;
    sqcfg)
      # instruction issue: what the waves of each kernel execute (per-wave VALU / SALU / memory instruction counts) and
      # how busy the issue ports are -- is a kernel bound by HBM or by the instruction stream?
      for c in ${TRACE_CONFIGS:-4}; do
        i=0
        for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" \
                 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
                 "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
          i=$((i+1))
          ( cd /tmp && timeout 900 rocprofv3 --pmc $C --output-format csv -d $O/sq_cfg$c/p$i -o pmc -- \
              env SLF_PLACEMENT_TUNE=0 python $R/tools/bench_configs.py --quick --only $c > $O/sq_cfg${c}_p$i.log 2>&1 )
        done
        python tools/pmc_summary.py $O/sq_cfg$c | tee $O/sq_summary_cfg$c.txt
      done ;;
    py)
      timeout ${PY_TIMEOUT:-900} python "$@" 2>&1 | tee -a $O/py.log | tail -${PY_TAIL:-40}; break ;;
    *) echo "unknown task $task" ;;
  esac
done
