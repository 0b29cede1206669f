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
  r5v2)   # what bounds the Shan-Chen kernels (SQ counters, kernel trace) + the x-slab step against RCCL's channel count
    TRACE_CONFIGS="4" bash tools/gpu.sh tracecfg sqcfg; rm -rf $O/trace_cfg*/ $O/sq_cfg*/
    X="--force_distributed --scaling strong --domain 128x512x512 --axis x --no_cpu_baseline --no_gpu_state --access_pattern AA --repeats 1"
    for ch in default 1 2 4; do
      if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
      timeout 300 python bench.py $X 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=c['per_rank'][0]; print('nchannels $ch', d['value'], d['best_value'], d['ms_per_step'], 'kernel', r['kernel_ms'], 'sweep_only', r['sweep_only_ms'], 'halo', r['halo_ms'], 'exposed', c['halo_exposed_ms'], c['validated'])" | tee -a $O/xslab_nchannels.txt
    done
    unset NCCL_MAX_NCHANNELS
    ;;
  r5v3)   # Shan-Chen after the potential became a compile-time constant of the psi blocks: parity, trace, SQ, PMC; x-slab vs RCCL channels
    ( time timeout 900 python -m pytest tests/test_gpu_sc.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not eight_subdomains" --durations=5 ) > $O/pytest_sc.log 2>&1; tail -12 $O/pytest_sc.log
    TRACE_CONFIGS="4" bash tools/gpu.sh tracecfg sqcfg pmccfg; rm -rf $O/trace_cfg*/ $O/sq_cfg*/ $O/pmc_cfg*/
    timeout 600 python tools/bench_configs.py --only 4 2>/dev/null | grep '^{' | tee $O/configs_sc.jsonl | cut -c1-200
    X="--gpus 1 --force_distributed --scaling strong --domain 128x512x512 --axis x --no_cpu_baseline --no_gpu_state --access_pattern AA --repeats 1"
    for ch in default 1 2 4; do
      if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
      timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py $X 2>&1 | tail -1 > $O/xslab_$ch.json
      python -c "import sys,json; d=json.load(open('$O/xslab_$ch.json')); c=d['config']; r=c['per_rank'][0]; print('nchannels $ch', d['value'], d['best_value'], d['ms_per_step'], 'kernel', r['kernel_ms'], 'sweep_only', r['sweep_only_ms'], 'halo', r['halo_ms'], 'exposed', c['halo_exposed_ms'], c['validated'], c['rccl_ranks'])" 2>&1 | tail -1 | tee -a $O/xslab_nchannels.txt
    done
    unset NCCL_MAX_NCHANNELS
    ;;
  r5v4)   # where the x-face buffers cost the slab's sweep (probe bits), and the x-slab step against MORE RCCL channels
    for pat in AA AB; do timeout 300 python tools/probe/xface_cost_probe.py $pat 40 7 2>&1 | grep -v amdgpu.ids | tee -a $O/xface_cost_probe.txt; done
    X="--gpus 1 --force_distributed --scaling strong --domain 128x512x512 --axis x --no_cpu_baseline --no_gpu_state --access_pattern AA --repeats 1"
    for ch in 8 16 32; do
      export NCCL_MIN_NCHANNELS=$ch
      timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py $X 2>&1 | tail -1 > $O/xslab_min$ch.json
      python -c "import sys,json; d=json.load(open('$O/xslab_min$ch.json')); c=d['config']; r=c['per_rank'][0]; print('min_nchannels $ch', d['value'], d['best_value'], d['ms_per_step'], 'kernel', r['kernel_ms'], 'sweep_only', r['sweep_only_ms'], 'halo', r['halo_ms'], 'exposed', c['halo_exposed_ms'], c['validated'], c['rccl_ranks'])" 2>&1 | tail -1 | tee -a $O/xslab_min_nchannels.txt
    done
    unset NCCL_MIN_NCHANNELS
    ;;
  r5v5)   # several steps per launch for small 2-D subdomains: parity, then BASELINE config 1 on the GPU
    ( time timeout 900 python -m pytest tests/test_gpu_resident.py -m gpu -q -x --durations=5 ) > $O/pytest_resident.log 2>&1; tail -25 $O/pytest_resident.log
    ( time timeout 600 python -m pytest tests/test_gpu_runner.py -m gpu -q -x -k "graphs or ldc_2d or poiseuille or checkpoint" ) > $O/pytest_runner_2d.log 2>&1; tail -5 $O/pytest_runner_2d.log
    timeout 300 python tools/bench_configs.py --only 0 2>/dev/null | grep '^{' | tee $O/configs_0_resident.jsonl | cut -c1-300
    SLF_RESIDENT=0 timeout 300 python tools/bench_configs.py --only 0 2>/dev/null | grep '^{' | tee $O/configs_0_stepping.jsonl | cut -c1-300
    for st in 4 6 10 12; do SLF_RESIDENT_STEPS=$st timeout 300 python tools/bench_configs.py --only 0 2>/dev/null | grep '^{' | tee $O/configs_0_resident_steps$st.jsonl | cut -c1-200; done
    ;;
  r5v6)   # what the resident kernel spends its time on: kernel trace + SQ counters of BASELINE config 1 on the GPU
    TRACE_CONFIGS="0" bash tools/gpu.sh tracecfg sqcfg; rm -rf $O/trace_cfg*/ $O/sq_cfg*/
    ;;
  r5v7)   # resident kernel with the active nodes compacted: parity, config 1 on the GPU against steps per launch, trace + SQ
    ( time timeout 900 python -m pytest tests/test_gpu_resident.py -m gpu -q -x --durations=3 ) > $O/pytest_resident.log 2>&1; tail -8 $O/pytest_resident.log
    for st in 8 10 12; do SLF_RESIDENT_STEPS=$st timeout 300 python tools/bench_configs.py --only 0 2>/dev/null | grep '^{' | tee $O/configs_0_resident_steps$st.jsonl | cut -c1-120; done
    TRACE_CONFIGS="0" bash tools/gpu.sh tracecfg sqcfg; rm -rf $O/trace_cfg*/ $O/sq_cfg*/
    ;;
  r5v8)   # resident kernel: mode-specialised steps + rectangle compaction; Shan-Chen with the potential as a template parameter
    ( time timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_sc.py -m gpu -q -x --durations=3 ) > $O/pytest_resident_sc.log 2>&1; tail -8 $O/pytest_resident_sc.log
    for st in 8 10 12; do SLF_RESIDENT_STEPS=$st timeout 300 python tools/bench_configs.py --only 0 2>/dev/null | grep '^{' | tee $O/configs_0_resident_steps$st.jsonl | cut -c1-120; done
    timeout 600 python tools/bench_configs.py --only 4 2>/dev/null | grep '^{' | tee $O/configs_sc.jsonl | cut -c1-120
    TRACE_CONFIGS="0 4" bash tools/gpu.sh tracecfg; TRACE_CONFIGS="0" bash tools/gpu.sh sqcfg; rm -rf $O/trace_cfg*/ $O/sq_cfg*/
    ;;
  r5v9)   # same-box A/B of the binary Shan-Chen kernels: round-4 sources / potential behind one branch / potential as template parameter
    for rep in 1 2; do
      for v in r04 branch template; do
        if [ $v = template ]; then unset SLF_LIBRARY; else export SLF_LIBRARY=$PWD/sailfish_amd/lib/libsailfish_hip_sc_$v.so; fi
        timeout 300 python tools/bench_configs.py --only 4 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['MLUPS_eff'], d['MLUPS_comp'])" | tee -a $O/sc_ab.txt
      done
    done
    unset SLF_LIBRARY
    ;;
  r5final)   # round-5 evidence visit: everything DESIGN.md / profiles/traffic.json quote for the shipped kernels
    bash tools/gpu.sh host smoke
    ( time timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > $O/pytest_gpu_final.log 2>&1; tail -22 $O/pytest_gpu_final.log
    timeout 900 python bench.py 2>&1 | tail -1 > $O/bench_final.json; cut -c1-400 $O/bench_final.json
    timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $O/bench_driver_cmd_final.json; cut -c1-300 $O/bench_driver_cmd_final.json
    for pat in AA AB; do
      BENCH_ARGS="--access_pattern $pat" bash tools/gpu.sh trace pmc; cp $O/kernel_stats.csv $O/kernel_stats_${pat}_final.csv; cp $O/pmc_summary.txt $O/pmc_summary_${pat}_512.txt; rm -rf $O/trace $O/pmc
    done
    timeout 1500 python tools/bench_configs.py 2>/dev/null | grep '^{' > $O/configs_final.jsonl; cut -c1-140 $O/configs_final.jsonl
    bash tools/gpu.sh torchrun; mv $O/torchrun.jsonl $O/torchrun_final.jsonl
    ;;
  r5v10)   # resident kernel with two nodes per lane in packed arithmetic: parity, config 1 on the GPU, trace + SQ
    ( time timeout 900 python -m pytest tests/test_gpu_resident.py -m gpu -q -x --durations=3 ) > $O/pytest_resident.log 2>&1; tail -8 $O/pytest_resident.log
    for st in 8 10 12; do SLF_RESIDENT_STEPS=$st timeout 300 python tools/bench_configs.py --only 0 2>/dev/null | grep '^{' | tee $O/configs_0_resident_steps$st.jsonl | cut -c1-120; done
    TRACE_CONFIGS="0" bash tools/gpu.sh tracecfg sqcfg; rm -rf $O/trace_cfg*/ $O/sq_cfg*/
    ;;
  r5final2)   # after the last test / doc changes: the tests that changed, the bench lines with `traffic` from the re-stamped file, f64
    ( time timeout 900 python -m pytest tests/test_gpu_face_kernels.py tests/test_gpu_comm.py tests/test_gpu_resident.py tests/test_gpu_sc.py -m gpu -q --durations=3 ) > $O/pytest_changed.log 2>&1; tail -8 $O/pytest_changed.log
    timeout 900 python bench.py 2>&1 | tail -1 > $O/bench_final.json; cut -c1-300 $O/bench_final.json
    timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $O/bench_driver_cmd_final.json; cut -c1-300 $O/bench_driver_cmd_final.json
    timeout 900 python bench.py --precision double --no_cpu_baseline 2>&1 | tail -1 > $O/bench_f64_final.json; cut -c1-300 $O/bench_f64_final.json
    timeout 900 python bench.py --model mrt --no_cpu_baseline 2>&1 | tail -1 > $O/bench_mrt_final.json; cut -c1-300 $O/bench_mrt_final.json
    ;;
  r5dn)   # do-nothing outlets (in place) and full-slip walls: their tests first, the headline line and the two PMC passes
          # profiles/traffic.json is stamped from (the kernel sources changed), then the whole GPU suite
    ( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py tests/test_gpu_resident.py -m gpu -q \
        -k "do_nothing or slip or outflow or boundary_condition_level or open_channel" --durations=5 ) > $O/pytest_open_nodes.log 2>&1; tail -12 $O/pytest_open_nodes.log
    for pat in AA AB; do
      PMC_SIZES_ONLY=1 BENCH_ARGS="--access_pattern $pat" bash tools/gpu.sh pmc; cp $O/pmc_summary.txt $O/pmc_summary_${pat}_512.txt
      [ $pat = AA ] && kern="slf::fast_" || kern="slf::fast_row_kernel"
      python tools/traffic_update.py --from-pmc $O/pmc --kernel "$kern" --key D3Q19_bgk_f32_${pat}_512_fused
      rm -rf $O/pmc
    done
    cp profiles/traffic.json $O/traffic.json
    timeout 900 python bench.py 2>&1 | tail -1 > $O/bench_final.json; cut -c1-400 $O/bench_final.json
    timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $O/bench_driver_cmd_final.json; cut -c1-300 $O/bench_driver_cmd_final.json
    ( time timeout 2400 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu_final.log 2>&1; tail -16 $O/pytest_gpu_final.log
    ;;
  r5last)   # the remaining evidence files on the final sources: smoke(), kernel traces of the headline command, f64 / MRT lines,
            # every single-GPU configuration
    bash tools/gpu.sh host smoke
    for pat in AA AB; do
      BENCH_ARGS="--access_pattern $pat" bash tools/gpu.sh trace; cp $O/kernel_stats.csv $O/kernel_stats_${pat}_final.csv; rm -rf $O/trace
    done
    timeout 900 python bench.py --precision double --no_cpu_baseline 2>&1 | tail -1 > $O/bench_f64_final.json; cut -c1-300 $O/bench_f64_final.json
    timeout 900 python bench.py --model mrt --no_cpu_baseline 2>&1 | tail -1 > $O/bench_mrt_final.json; cut -c1-300 $O/bench_mrt_final.json
    timeout 600 python tools/bench_configs.py 2>/dev/null | grep '^{' > $O/configs_final.jsonl; cut -c1-140 $O/configs_final.jsonl
    ;;
  r5duct)   # what the level-2 kernels cost where EVERY row holds boundary nodes: an open duct along x against one along z
    timeout 500 python tools/bench_configs.py --only 6xa,6xb,6za,6zb 2>/dev/null | grep '^{' | tee $O/configs_open_duct.jsonl | cut -c1-200
    ;;
  r5bcw)   # A/B on one box: the boundary-condition instantiations of the whole-row kernels asked for 6 (ships), 7 or 8 resident
           # waves per SIMD (sailfish_amd/lib/variants/, built with -DSLF_BC_WAVES=N), on the duct whose every row holds boundary nodes
    for round in 1 2; do
      for lib in "" sailfish_amd/lib/variants/libsailfish_hip_w7.so sailfish_amd/lib/variants/libsailfish_hip_w8.so; do
        echo "== library: ${lib:-shipped}" | tee -a $O/bc_waves_ab.txt
        SLF_LIBRARY=${lib:+$PWD/$lib} timeout 200 python tools/bench_configs.py --only 6xa,6xb 2>/dev/null | grep '^{' | cut -c1-150 | tee -a $O/bc_waves_ab.txt
      done
    done
    ;;
  r5bcw2)   # the same A/B, shipped against 7 waves only, full-length runs (1500 steps each)
    for round in 1 2; do
      for lib in "" sailfish_amd/lib/variants/libsailfish_hip_w7.so; do
        echo "== library: ${lib:-shipped}" | tee -a $O/bc_waves_ab_long.txt
        SLF_LIBRARY=${lib:+$PWD/$lib} timeout 200 python tools/bench_configs.py --only 6xa,6xb 2>/dev/null | grep '^{' | cut -c1-150 | tee -a $O/bc_waves_ab_long.txt
      done
    done
    ;;
  r5final3)   # the whole GPU suite on the final tree
    ( time timeout 2400 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu_final.log 2>&1; tail -16 $O/pytest_gpu_final.log
    ;;
  r5v11)   # config 4's eight subdomains in ONE process on the one GPU (LocalGroup)
    timeout 900 python tools/bench_configs.py --only 3g8 2>&1 | tail -5 | tee $O/configs_3g8.jsonl | cut -c1-400
    ;;
  r5v12)   # an example script that starts eight ranks itself
    ( time timeout 900 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q -x -k "starts_its_own_ranks" --durations=4 ) > $O/pytest_own_ranks.log 2>&1; tail -14 $O/pytest_own_ranks.log
    ;;
  r5v13)   # reference-style command lines at realistic sizes with several subdomains in one process (resource checks)
    for sub in "2 z AA" "4 x AB"; do
      set -- $sub
      echo "== binary Shan-Chen 256^3, $1 subdomains along $2, $3" | tee -a $O/multi_subdomain_runs.txt
      ( timeout 600 python -c "
import sys; sys.path.insert(0, '.')
from examples.binary_fluid.sc_separation_3d import SeparationSim
from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
c = LBSimulationController(SeparationSim, EqualSubdomainsGeometry3D, default_config=dict(lat_nx=256, lat_ny=256, lat_nz=256, subdomains=$1, conn_axis='$2', access_pattern='$3', mode='benchmark', max_iters=300, benchmark_sample_from=100, perf_stats_every=0))
c.run(ignore_cmdline=True)
" 2>&1 | grep -v amdgpu.ids | tail -4 ) | tee -a $O/multi_subdomain_runs.txt
    done
    for cmd in \
               "examples/ldc_3d.py --lat_nx=512 --lat_ny=512 --lat_nz=512 --model=mrt --subdomains=4 --conn_axis=x --access_pattern=AA --visc=0.0256" \
               "examples/ldc_3d.py --lat_nx=512 --lat_ny=512 --lat_nz=512 --subdomains=8 --conn_axis=z --access_pattern=AB --visc=0.0256" \
               "examples/poiseuille_3d.py --lat_nx=512 --lat_ny=256 --lat_nz=256 --subdomains=3 --conn_axis=x --visc=0.05 --access_pattern=AA" ; do
      echo "== $cmd" | tee -a $O/multi_subdomain_runs.txt
      ( timeout 600 python $cmd --mode=benchmark --max_iters=300 --benchmark_sample_from=100 --perf_stats_every=0 2>&1 | grep -v amdgpu.ids | tail -4 ) | tee -a $O/multi_subdomain_runs.txt
    done
    ;;
  r5v14)   # where a pipe cut into three x-slabs in one process loses a third of its rate: kernel trace
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o trace -- \
        env SLF_PLACEMENT_TUNE=0 python $GRAFT_REPO_ROOT/examples/poiseuille_3d.py --lat_nx=512 --lat_ny=256 --lat_nz=256 --subdomains=3 --conn_axis=x --visc=0.05 --access_pattern=AA --mode=benchmark --max_iters=300 --benchmark_sample_from=100 --perf_stats_every=0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 )
    for f in $(find $O/trace -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_pipe_3x.csv; head -14 $f | cut -c1-220; done; rm -rf $O/trace; tail -3 $O/trace.log
    ;;
  r5v15)   # x-face buffers without the per-step clear and shared between the subdomains of a process: parity, then the rates
    ( time timeout 1500 python -m pytest tests/test_gpu_runner.py tests/test_gpu_slab.py tests/test_gpu_kat.py tests/test_gpu_two_ranks.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not eight_subdomains" --durations=4 ) > $O/pytest_xface.log 2>&1; tail -10 $O/pytest_xface.log
    timeout 900 python tools/bench_configs.py --only 3,3b,3g8 2>/dev/null | grep '^{' | tee $O/configs_xslabs_shared.jsonl | cut -c1-170
    SLF_XFACE_SHARE=0 timeout 900 python tools/bench_configs.py --only 3,3b,3g8 2>/dev/null | grep '^{' | tee $O/configs_xslabs_copied.jsonl | cut -c1-170
    for v in "SLF_XFACE_SHARE=1 SLF_XFACE_CLEAR=0" "SLF_XFACE_SHARE=0 SLF_XFACE_CLEAR=0" "SLF_XFACE_SHARE=0 SLF_XFACE_CLEAR=1"; do
      echo "== pipe 512x256x256 in 3 x-slabs, one process, $v" | tee -a $O/pipe_3x.txt
      ( env $v timeout 600 python examples/poiseuille_3d.py --lat_nx=512 --lat_ny=256 --lat_nz=256 --subdomains=3 --conn_axis=x --visc=0.05 --access_pattern=AA --mode=benchmark --max_iters=300 --benchmark_sample_from=100 --perf_stats_every=0 2>&1 | grep "Total MLUPS" ) | tee -a $O/pipe_3x.txt
    done
    ;;
  r5v16)   # the x-slab pair with shared / copied face buffers, alternating on one box
    for rep in 1 2 3; do
      for sh in 1 0; do
        SLF_XFACE_SHARE=$sh timeout 600 python tools/bench_configs.py --only 3 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('share=$sh', d['MLUPS_eff'], d['MLUPS_comp'])" | tee -a $O/pair_share_ab.txt
      done
    done
    ;;
  r5v17)   # group fast-forward: parity of every multi-subdomain test, then rates with / without
    ( time timeout 1500 python -m pytest tests/test_gpu_runner.py tests/test_gpu_sc.py tests/test_gpu_examples.py -m gpu -q -x --durations=4 ) > $O/pytest_groups.log 2>&1; tail -8 $O/pytest_groups.log
    for ff in 1 0; do
      echo "== SLF_GROUP_FAST_FORWARD=$ff" | tee -a $O/group_ff.txt
      SLF_GROUP_FAST_FORWARD=$ff timeout 900 python tools/bench_configs.py --only 3,3g8 2>/dev/null | grep '^{' | cut -c1-150 | tee -a $O/group_ff.txt
      ( SLF_GROUP_FAST_FORWARD=$ff timeout 600 python examples/poiseuille_3d.py --lat_nx=512 --lat_ny=256 --lat_nz=256 --subdomains=3 --conn_axis=x --visc=0.05 --access_pattern=AA --mode=benchmark --max_iters=300 --benchmark_sample_from=100 --perf_stats_every=0 2>&1 | grep "Total MLUPS" ) | tee -a $O/group_ff.txt
      ( SLF_GROUP_FAST_FORWARD=$ff timeout 600 python examples/ldc_3d.py --lat_nx=256 --lat_ny=256 --lat_nz=256 --subdomains=4 --conn_axis=z --visc=0.03 --access_pattern=AA --mode=benchmark --max_iters=600 --benchmark_sample_from=200 --perf_stats_every=0 2>&1 | grep "Total MLUPS" ) | tee -a $O/group_ff.txt
    done
    ;;
  r5v18)   # the 256^2 cavity in the two-copy pattern (the reference's default): one launch per step against 7 steps per launch
    for res in 1 0; do
      ( SLF_RESIDENT=$res timeout 300 python examples/ldc_2d.py --lat_nx=256 --lat_ny=256 --visc=0.0254 --access_pattern=AB --mode=benchmark --max_iters=40000 --benchmark_sample_from=10000 --perf_stats_every=0 2>&1 | grep "Total MLUPS" | sed "s/^/AB resident=$res /" ) | tee -a $O/ldc2d_ab.txt
      ( SLF_RESIDENT=$res timeout 300 python examples/ldc_2d.py --lat_nx=256 --lat_ny=256 --visc=0.0254 --access_pattern=AA --model=mrt --mode=benchmark --max_iters=40000 --benchmark_sample_from=10000 --perf_stats_every=0 2>&1 | grep "Total MLUPS" | sed "s/^/AA MRT resident=$res /" ) | tee -a $O/ldc2d_ab.txt
      ( SLF_RESIDENT=$res timeout 300 python examples/ldc_2d.py --lat_nx=256 --lat_ny=256 --visc=0.0254 --access_pattern=AA --precision=double --mode=benchmark --max_iters=40000 --benchmark_sample_from=10000 --perf_stats_every=0 2>&1 | grep "Total MLUPS" | sed "s/^/AA f64 resident=$res /" ) | tee -a $O/ldc2d_ab.txt
      ( SLF_RESIDENT=$res timeout 300 python examples/ldc_2d.py --lat_nx=128 --lat_ny=128 --visc=0.0254 --access_pattern=AA --mode=benchmark --max_iters=40000 --benchmark_sample_from=10000 --perf_stats_every=0 2>&1 | grep "Total MLUPS" | sed "s/^/AA 128^2 resident=$res /" ) | tee -a $O/ldc2d_ab.txt
      ( SLF_RESIDENT=$res timeout 300 python examples/ldc_2d.py --lat_nx=512 --lat_ny=512 --visc=0.0254 --access_pattern=AA --mode=benchmark --max_iters=20000 --benchmark_sample_from=5000 --perf_stats_every=0 2>&1 | grep "Total MLUPS" | sed "s/^/AA 512^2 resident=$res /" ) | tee -a $O/ldc2d_ab.txt
    done
    ;;
  r5v19)   # the resident path's eligibility rule
    ( time timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_runner.py -m gpu -q -x --durations=3 ) > $O/pytest_resident_rule.log 2>&1; tail -8 $O/pytest_resident_rule.log
    ;;
  r6v1)   # peer transport (HIP IPC mappings + progress counters): one rank against itself, then 2 / 8 processes on the one GPU
    export SLF_PEER_TIMEOUT_S=30
    X="--steps 40 --warmup 10 --prewarm_steps 40 --repeats 1 --no_cpu_baseline --no_gpu_state --min_seconds 0.3 --halo_timing_steps 6"
    line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('config',{}); r=(c.get('per_rank') or [{}])[0]; print('$1', d.get('value'), d.get('ms_per_step'), c.get('access_pattern'), 'validated', c.get('validated'), 'rccl_ranks', c.get('rccl_ranks'), 'exposed', c.get('halo_exposed_ms'), 'sweep_only', r.get('sweep_only_ms'), '|', (c.get('halo_transport') or '')[:60], d.get('error'))" 2>&1 | tail -1; }
    for tr in peer rccl; do
      for ax in x z; do
        SLF_HALO_TRANSPORT=$tr timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
          bench.py --gpus 1 --force_distributed --scaling strong --domain 128x512x512 --axis $ax $X 2> $O/one_${tr}_$ax.err | tail -1 > $O/one_${tr}_$ax.json
        line "one rank $tr $ax" < $O/one_${tr}_$ax.json | tee -a $O/summary.txt
      done
    done
    for n in 2 8; do
      for ax in x z; do
        SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 900 python bench.py --gpus $n --scaling strong --domain 1024x512x512 --axis $ax $X 2> $O/n${n}_$ax.err | tail -1 > $O/n${n}_$ax.json
        line "$n processes $ax" < $O/n${n}_$ax.json | tee -a $O/summary.txt
      done
    done
    tail -n 5 $O/*.err | tail -n 60
    ;;
  r6v2)   # eight processes on the one GPU: hardware queues per process x z-chunks of the x-split sweep
    export SLF_PEER_TIMEOUT_S=30
    X="--steps 40 --warmup 10 --prewarm_steps 40 --repeats 1 --no_cpu_baseline --no_gpu_state --min_seconds 0.3 --halo_timing_steps 6 --access_pattern AA --no_validate"
    line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('config',{}); r=(c.get('per_rank') or [{}])[0]; print('$1', d.get('value'), d.get('ms_per_step'), c.get('access_pattern'), 'validated', c.get('validated'), 'exposed', c.get('halo_exposed_ms'), 'sweep_only', r.get('sweep_only_ms'), d.get('error'))" 2>&1 | tail -1; }
    for q in 3 4; do GPU_MAX_HW_QUEUES=$q timeout 120 tools/probe/ipc_probe 8 262144 10 256 2>&1 | grep "ring ping" | sed "s/^/queues $q: /" | tee -a $O/summary.txt; done
    for q in 2 3 4; do
      for ch in 4 2 1; do
        GPU_MAX_HW_QUEUES=$q SLF_XFACE_CHUNKS=$ch SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 600 python bench.py --gpus 8 --scaling strong --domain 1024x512x512 --axis x $X 2> $O/n8_x_q${q}_c$ch.err | tail -1 > $O/n8_x_q${q}_c$ch.json
        line "8 processes x, $q queues, $ch chunks" < $O/n8_x_q${q}_c$ch.json | tee -a $O/summary.txt
      done
    done
    for q in 3 4; do
      GPU_MAX_HW_QUEUES=$q SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 600 python bench.py --gpus 8 --scaling strong --domain 1024x512x512 --axis z $X 2> $O/n8_z_q$q.err | tail -1 > $O/n8_z_q$q.json
      line "8 processes z, $q queues" < $O/n8_z_q$q.json | tee -a $O/summary.txt
    done
    ;;
  r6v3)   # peer transport, waits in front of the chunks that need them: parity tests, then 1 / 2 / 8 processes
    export SLF_PEER_TIMEOUT_S=30
    ( time timeout 1500 python -m pytest tests/test_gpu_peer.py tests/test_gpu_two_ranks.py tests/test_gpu_slab.py tests/test_gpu_comm.py -m gpu -q -x --durations=8 ) > $O/pytest_peer.log 2>&1; tail -25 $O/pytest_peer.log
    X="--steps 40 --warmup 10 --prewarm_steps 40 --repeats 1 --no_cpu_baseline --no_gpu_state --min_seconds 0.3 --halo_timing_steps 6"
    line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('config',{}); r=(c.get('per_rank') or [{}])[0]; print('$1', d.get('value'), d.get('ms_per_step'), c.get('access_pattern'), 'validated', c.get('validated'), 'exposed', c.get('halo_exposed_ms'), 'sweep_only', r.get('sweep_only_ms'), '|', (c.get('halo_transport') or '')[:24], d.get('error'))" 2>&1 | tail -1; }
    for tr in peer rccl; do
      SLF_HALO_TRANSPORT=$tr timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
        bench.py --gpus 1 --force_distributed --scaling strong --domain 128x512x512 --axis x $X 2> $O/one_${tr}_x.err | tail -1 > $O/one_${tr}_x.json
      line "one rank $tr x" < $O/one_${tr}_x.json | tee -a $O/summary.txt
    done
    for ch in 4 2 1; do
      SLF_XFACE_CHUNKS=$ch SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 8 --scaling strong --domain 1024x512x512 --axis x $X 2> $O/n8_x_c$ch.err | tail -1 > $O/n8_x_c$ch.json
      line "8 processes x, $ch chunks" < $O/n8_x_c$ch.json | tee -a $O/summary.txt
    done
    for v in "SLF_HALO_PRIORITY=1 SLF_CALC_STREAMS=2" "SLF_HALO_PRIORITY=0 SLF_CALC_STREAMS=2" "SLF_HALO_PRIORITY=1 SLF_CALC_STREAMS=1"; do
      env $v SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 8 --scaling strong --domain 1024x512x512 --axis z $X --no_validate --access_pattern AA 2> $O/n8_z.err | tail -1 > $O/n8_z.json
      line "8 processes z, $v" < $O/n8_z.json | tee -a $O/summary.txt
    done
    SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 2 --scaling strong --domain 1024x512x512 --axis x $X 2> $O/n2_x.err | tail -1 > $O/n2_x.json
    line "2 processes x" < $O/n2_x.json | tee -a $O/summary.txt
    tail -n 5 $O/*.err | tail -n 40
    ;;
  r6v4)   # peer transport with deferred waits on both paths, no stream priority between processes that share the device
    export SLF_PEER_TIMEOUT_S=30
    ( time timeout 1500 python -m pytest tests/test_gpu_peer.py tests/test_gpu_two_ranks.py tests/test_gpu_slab.py tests/test_gpu_comm.py -m gpu -q --durations=8 ) > $O/pytest_peer.log 2>&1; tail -25 $O/pytest_peer.log
    X="--steps 40 --warmup 10 --prewarm_steps 40 --repeats 1 --no_cpu_baseline --no_gpu_state --min_seconds 0.3 --halo_timing_steps 6"
    line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('config',{}); r=(c.get('per_rank') or [{}])[0]; print('$1', d.get('value'), d.get('ms_per_step'), c.get('access_pattern'), 'validated', c.get('validated'), 'exposed', c.get('halo_exposed_ms'), 'sweep_only', r.get('sweep_only_ms'), '|', (c.get('halo_transport') or '')[:24], d.get('error'))" 2>&1 | tail -1; }
    for tr in peer rccl; do
      for ax in x z; do
        SLF_HALO_TRANSPORT=$tr timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
          bench.py --gpus 1 --force_distributed --scaling strong --domain 128x512x512 --axis $ax $X 2> $O/one_${tr}_$ax.err | tail -1 > $O/one_${tr}_$ax.json
        line "one rank $tr $ax" < $O/one_${tr}_$ax.json | tee -a $O/summary.txt
      done
    done
    for ch in 4 2 1; do
      SLF_XFACE_CHUNKS=$ch SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 8 --scaling strong --domain 1024x512x512 --axis x $X 2> $O/n8_x_c$ch.err | tail -1 > $O/n8_x_c$ch.json
      line "8 processes x, $ch chunks" < $O/n8_x_c$ch.json | tee -a $O/summary.txt
    done
    for v in "SLF_CALC_STREAMS=2" "SLF_CALC_STREAMS=1"; do
      env $v SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 8 --scaling strong --domain 1024x512x512 --axis z $X 2> $O/n8_z.err | tail -1 > $O/n8_z_$v.json
      line "8 processes z, $v" < $O/n8_z_$v.json | tee -a $O/summary.txt
    done
    for ax in x z; do
      SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 2 --scaling strong --domain 1024x512x512 --axis $ax $X 2> $O/n2_$ax.err | tail -1 > $O/n2_$ax.json
      line "2 processes $ax" < $O/n2_$ax.json | tee -a $O/summary.txt
    done
    grep -v "socket.cpp\|amdgpu.ids" $O/*.err | tail -n 30
    ;;
  r6v5)   # the AB failures of r6v4: the second simulation of a process, or the two-copy pattern?  + the failed tests again
    export SLF_PEER_TIMEOUT_S=30
    X="--steps 40 --warmup 10 --prewarm_steps 40 --repeats 1 --no_cpu_baseline --no_gpu_state --min_seconds 0.3 --halo_timing_steps 6"
    line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('config',{}); r=(c.get('per_rank') or [{}])[0]; print('$1', d.get('value'), d.get('ms_per_step'), c.get('access_pattern'), 'validated', c.get('validated'), dict((k, v.get('populations_bit_identical')) for k, v in c.get('validation', {}).items()), 'exposed', c.get('halo_exposed_ms'), '|', (c.get('halo_transport') or '')[:24], d.get('error'))" 2>&1 | tail -1; }
    for ax in x z; do
      for pat in AB auto; do
        SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 2 --scaling strong --domain 512x256x256 --axis $ax --access_pattern $pat $X 2> $O/n2_${ax}_$pat.err | tail -1 > $O/n2_${ax}_$pat.json
        line "2 processes $ax $pat" < $O/n2_${ax}_$pat.json | tee -a $O/summary.txt
      done
    done
    ( time timeout 1500 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q --durations=8 -k "stops or controller_branch or equal_one_box" ) > $O/pytest_peer.log 2>&1; tail -12 $O/pytest_peer.log
    for ax in x z; do
      SLF_DIST_BACKEND=gloo SLF_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 8 --scaling strong --domain 1024x512x512 --axis $ax $X 2> $O/n8_$ax.err | tail -1 > $O/n8_$ax.json
      line "8 processes $ax" < $O/n8_$ax.json | tee -a $O/summary.txt
    done
    grep -v "socket.cpp\|amdgpu.ids" $O/*.err | tail -n 30
    ;;
  r6v6)   # the whole GPU suite after the peer transport + the rows of SURVEY 8(f) that had no number (indirect addressing,
          # single-component Shan-Chen, obstacle examples) + the open ducts of round 5
    export SLF_PEER_TIMEOUT_S=30
    ( time timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > $O/pytest_gpu.log 2>&1; tail -22 $O/pytest_gpu.log
    for c in 7a 7b 7c 7d 8a 8b 9a 9b 6xa 6xb; do
      timeout 600 python tools/bench_configs.py --only $c 2> $O/cfg_$c.err | grep '^{' | tee -a $O/configs_new.jsonl | cut -c1-330
    done
    TRACE_CONFIGS="7a 8a 8b" bash tools/gpu.sh tracecfg pmccfg; rm -rf $O/trace_cfg*/ $O/pmc_cfg*/
    grep -v "amdgpu.ids" $O/cfg_*.err | tail -n 30
    ;;
  r6v7)   # split rows: parity, then the open ducts with / without (alternating on this box); where several subdomains in
          # one process lose their rate: kernel traces of the pipe in 3 x-slabs and of binary Shan-Chen in 4 x-slabs
    ( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py -m gpu -q -x -k "split or row_classes or open_channel or do_nothing or duct or cylinder or sphere" --durations=5 ) > $O/pytest_split.log 2>&1; tail -8 $O/pytest_split.log
    for rep in 1 2; do
      for sp in 1 0; do
        for c in 6xa 6xb; do
          SLF_ROW_SPLIT=$sp timeout 600 python tools/bench_configs.py --only $c 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $sp', d['config'][:4], d['MLUPS_eff'], d['MLUPS_comp'])" | tee -a $O/split_rows_ab.txt
        done
      done
    done
    P="--lat_nx=512 --lat_ny=256 --lat_nz=256 --subdomains=3 --conn_axis=x --visc=0.05 --access_pattern=AA --mode=benchmark --max_iters=300 --benchmark_sample_from=100 --perf_stats_every=0"
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_pipe -o trace -- \
        env SLF_PLACEMENT_TUNE=0 python $GRAFT_REPO_ROOT/examples/poiseuille_3d.py $P > $GRAFT_REPO_ROOT/$O/trace_pipe.log 2>&1 )
    python tools/probe/trace_busy.py $O/trace_pipe | tee $O/trace_busy_pipe_3x.txt; rm -rf $O/trace_pipe; grep "Total MLUPS" $O/trace_pipe.log
    cat > /tmp/sc4.py <<PYEOF
import sys
sys.path.insert(0, '$GRAFT_REPO_ROOT')
from examples.binary_fluid.sc_separation_3d import SeparationSim
from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
c = LBSimulationController(SeparationSim, EqualSubdomainsGeometry3D, default_config=dict(lat_nx=256, lat_ny=256, lat_nz=256, subdomains=4, conn_axis='x', access_pattern='AB', mode='benchmark', max_iters=300, benchmark_sample_from=100, perf_stats_every=0))
c.run(ignore_cmdline=True)
PYEOF
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_sc -o trace -- \
        env SLF_PLACEMENT_TUNE=0 python /tmp/sc4.py > $GRAFT_REPO_ROOT/$O/trace_sc.log 2>&1 )
    python tools/probe/trace_busy.py $O/trace_sc | tee $O/trace_busy_sc_4x.txt; rm -rf $O/trace_sc; grep "Total MLUPS" $O/trace_sc.log
    timeout 300 python /tmp/sc4.py 2>&1 | grep "Total MLUPS" | tee $O/sc_4x_untraced.txt
    timeout 300 python examples/poiseuille_3d.py $P 2>&1 | grep "Total MLUPS" | tee $O/pipe_3x_untraced.txt
    ;;
  r6v8)   # indirect addressing, one thread per slot: parity, then the packed bed and the wavy pipe with / without;
          # subdomains of one process with their sweeps on one stream: parity, pipe in 3 x-slabs, config 4's slabs in one process
    export SLF_PEER_TIMEOUT_S=30
    ( time timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py tests/test_gpu_sc.py tests/test_gpu_kat.py tests/test_gpu_examples.py -m gpu -q -x --durations=5 ) > $O/pytest_indirect_groups.log 2>&1; tail -8 $O/pytest_indirect_groups.log
    ( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "eight_subdomains and z-peer" ) > $O/pytest_zpeer.log 2>&1; tail -4 $O/pytest_zpeer.log
    for sl in 1 0; do
      for c in 7a 7c; do
        SLF_INDIRECT_SLOTS=$sl timeout 600 python tools/bench_configs.py --only $c 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slots $sl', d['config'][:3], d['MLUPS_eff'], d['MLUPS_comp'], d.get('frac_of_8TBps'))" | tee -a $O/indirect_slots_ab.txt
      done
    done
    TRACE_CONFIGS="7a" bash tools/gpu.sh tracecfg pmccfg; rm -rf $O/trace_cfg*/ $O/pmc_cfg*/
    P="--lat_nx=512 --lat_ny=256 --lat_nz=256 --subdomains=3 --conn_axis=x --visc=0.05 --access_pattern=AA --mode=benchmark --max_iters=600 --benchmark_sample_from=200 --perf_stats_every=0"
    for one in 1 0 1 0; do
      ( SLF_GROUP_ONE_STREAM=$one timeout 300 python examples/poiseuille_3d.py $P 2>&1 | grep "Total MLUPS" | sed "s/^/one stream $one: pipe 3 x-slabs /" ) | tee -a $O/group_one_stream_ab.txt
    done
    for one in 1 0; do
      SLF_GROUP_ONE_STREAM=$one timeout 900 python tools/bench_configs.py --only 3,3b,3g8 2>/dev/null | grep '^{' | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print('one stream $one', d['config'][:4], d['MLUPS_eff'], d.get('frac_of_8TBps'))" | tee -a $O/group_one_stream_ab.txt
    done
    ;;
  r6v9)   # indirect addressing over the slots (table built when the arguments are bound), --regularized / --subgrid
    export SLF_PEER_TIMEOUT_S=30
    ( time timeout 1800 python -m pytest tests/test_gpu_reg_les.py tests/test_gpu_parity.py tests/test_gpu_runner.py tests/test_gpu_sc.py tests/test_gpu_golden.py tests/test_gpu_resident.py -m gpu -q -x --durations=5 ) > $O/pytest_reg_les_indirect.log 2>&1; tail -8 $O/pytest_reg_les_indirect.log
    for sl in 1 0 1 0; do
      for c in 7a 7c; do
        SLF_INDIRECT_SLOTS=$sl timeout 600 python tools/bench_configs.py --only $c 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slots $sl', d['config'][:3], d['MLUPS_eff'], d['MLUPS_comp'], d.get('frac_of_8TBps'))" | tee -a $O/indirect_slots_ab.txt
      done
    done
    TRACE_CONFIGS="7a" bash tools/gpu.sh tracecfg pmccfg; rm -rf $O/trace_cfg*/ $O/pmc_cfg*/
    ;;
  r6s1)   # binary Shan-Chen over x-face planes: parity, processes, then 256^3 in 4 x-slabs / 2 x-slabs / undivided
    ( time timeout 1500 python -m pytest tests/test_gpu_sc.py -m gpu -q -x -k "planes or multi_subdomain or checkpoint" --durations=5 ) > $O/pytest_sc_planes.log 2>&1; tail -15 $O/pytest_sc_planes.log
    ( time timeout 900 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q -x -k "shan_chen" --durations=5 ) > $O/pytest_sc_ranks.log 2>&1; tail -8 $O/pytest_sc_ranks.log
    cat > /tmp/sc4.py <<PYEOF
import sys
sys.path.insert(0, '$GRAFT_REPO_ROOT')
from examples.binary_fluid.sc_separation_3d import SeparationSim
from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
n = int(sys.argv[1])
c = LBSimulationController(SeparationSim, EqualSubdomainsGeometry3D, default_config=dict(lat_nx=256, lat_ny=256, lat_nz=256, subdomains=n, conn_axis='x', access_pattern='AB', mode='benchmark', max_iters=300, benchmark_sample_from=100, perf_stats_every=0))
c.run(ignore_cmdline=True)
PYEOF
    for rep in 1 2; do
      for n in 4 2 1; do
        for xf in 1 0; do
          echo "planes $xf subdomains $n: $(SLF_SC_XFACE=$xf timeout 300 python /tmp/sc4.py $n 2>&1 | grep 'Total MLUPS')" | tee -a $O/sc_x_slabs_ab.txt
        done
      done
    done
    echo "planes 1 subdomains 4, one stream 0: $(SLF_GROUP_ONE_STREAM=0 timeout 300 python /tmp/sc4.py 4 2>&1 | grep 'Total MLUPS')" | tee -a $O/sc_x_slabs_ab.txt
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_sc -o trace -- \
        env SLF_PLACEMENT_TUNE=0 python /tmp/sc4.py 4 > $GRAFT_REPO_ROOT/$O/trace_sc.log 2>&1 )
    python tools/probe/trace_busy.py $O/trace_sc | tee $O/trace_busy_sc_4x_planes.txt; rm -rf $O/trace_sc
    ;;
  r6s2)   # x-slab cuts on 128-byte lines: the force-driven pipe 512x256x256 in 3 x-slabs, 170/170/172 against 160/192/160
    P="--lat_nx=512 --lat_ny=256 --lat_nz=256 --subdomains=3 --conn_axis=x --visc=0.05 --access_pattern=AA --mode=benchmark --max_iters=600 --benchmark_sample_from=200 --perf_stats_every=0"
    for rep in 1 2; do
      for al in 0 32; do
        echo "pipe AA slab_align $al: $(timeout 300 python examples/poiseuille_3d.py $P --slab_align=$al 2>&1 | grep 'Total MLUPS')" | tee -a $O/slab_align_ab.txt
      done
    done
    for al in 0 32; do
      echo "pipe AB slab_align $al: $(timeout 300 python examples/poiseuille_3d.py ${P/=AA/=AB} --slab_align=$al 2>&1 | grep 'Total MLUPS')" | tee -a $O/slab_align_ab.txt
    done
    echo "pipe AA undivided: $(timeout 300 python examples/poiseuille_3d.py ${P/subdomains=3/subdomains=1} 2>&1 | grep 'Total MLUPS')" | tee -a $O/slab_align_ab.txt
    ( time timeout 1200 python -m pytest tests/test_gpu_runner.py tests/test_gpu_slab.py -m gpu -q -x -k "subdomain or slab or group" --durations=5 ) > $O/pytest_slabs.log 2>&1; tail -6 $O/pytest_slabs.log
    ;;
  r6pmc)   # the PMC passes of the headline kernels + bench lines for the sources as they are (after a kernel-source change)
    for pat in AA AB; do
      PMC_SIZES_ONLY=1 BENCH_ARGS="--access_pattern $pat" bash tools/gpu.sh pmc; cp $O/pmc_summary.txt $O/pmc_sizes_${pat}_512_final.txt
      [ $pat = AA ] && kern="slf::fast_" || kern="slf::fast_row_kernel"
      python tools/traffic_update.py --from-pmc $O/pmc --kernel "$kern" --key D3Q19_bgk_f32_${pat}_512_fused
      rm -rf $O/pmc
    done
    cp profiles/traffic.json $O/traffic.json
    timeout 900 python bench.py 2>&1 | tail -1 > $O/bench_final.json; cut -c1-400 $O/bench_final.json
    timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $O/bench_driver_cmd_final.json; cut -c1-300 $O/bench_driver_cmd_final.json
    ;;
  r6s3)   # the planes in the in-place pattern: parity, checkpoints, then 256^3 in 4 x-slabs AA with / without
    ( time timeout 1500 python -m pytest tests/test_gpu_sc.py -m gpu -q -x --durations=5 ) > $O/pytest_sc_all.log 2>&1; tail -12 $O/pytest_sc_all.log
    ( time timeout 900 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q -x -k "shan_chen" --durations=5 ) > $O/pytest_sc_ranks.log 2>&1; tail -5 $O/pytest_sc_ranks.log
    cat > /tmp/sc4.py <<PYEOF
import sys
sys.path.insert(0, '$GRAFT_REPO_ROOT')
from examples.binary_fluid.sc_separation_3d import SeparationSim
from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
n = int(sys.argv[1])
c = LBSimulationController(SeparationSim, EqualSubdomainsGeometry3D, default_config=dict(lat_nx=256, lat_ny=256, lat_nz=256, subdomains=n, conn_axis='x', access_pattern=sys.argv[2], mode='benchmark', max_iters=300, benchmark_sample_from=100, perf_stats_every=0))
c.run(ignore_cmdline=True)
PYEOF
    for rep in 1 2; do
      for n in 4 1; do
        for xf in 1 0; do
          echo "AA planes $xf subdomains $n: $(SLF_SC_XFACE=$xf timeout 300 python /tmp/sc4.py $n AA 2>&1 | grep 'Total MLUPS')" | tee -a $O/sc_x_slabs_aa.txt
        done
      done
    done
    ;;
  r6s4)   # single-component Shan-Chen over the planes: parity; 256^3 in 4 x-slabs with / without
    ( time timeout 1500 python -m pytest tests/test_gpu_sc.py -m gpu -q -x --durations=5 ) > $O/pytest_sc_all.log 2>&1; tail -12 $O/pytest_sc_all.log
    ( time timeout 900 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q -x -k "shan_chen" --durations=5 ) > $O/pytest_sc_ranks.log 2>&1; tail -5 $O/pytest_sc_ranks.log
    cat > /tmp/scs4.py <<PYEOF
import sys
sys.path.insert(0, '$GRAFT_REPO_ROOT')
sys.path.insert(0, '$GRAFT_REPO_ROOT/tools')
from examples.sc_phase_separation import PhaseSeparationSim, VapourSubdomain
from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
from sailfish.subdomain import Subdomain3D
import numpy as np
class Vapour3D(Subdomain3D):
    def boundary_conditions(self, hx, hy, hz):
        pass
    def initial_conditions(self, sim, hx, hy, hz):
        sim.rho[:] = 0.693 + 0.01 * np.sin(hx * 12.9898 + hy * 78.233 + hz * 37.719)
class Sim3D(PhaseSeparationSim):
    subdomain = Vapour3D
n = int(sys.argv[1])
c = LBSimulationController(Sim3D, EqualSubdomainsGeometry3D, default_config=dict(lat_nx=256, lat_ny=256, lat_nz=256, grid='D3Q19', periodic_x=True, periodic_y=True, periodic_z=True, subdomains=n, conn_axis='x', access_pattern=sys.argv[2], mode='benchmark', max_iters=300, benchmark_sample_from=100, perf_stats_every=0))
c.run(ignore_cmdline=True)
PYEOF
    for pat in AA AB; do
      for n in 4 1; do
        for xf in 1 0; do
          echo "single-component $pat planes $xf subdomains $n: $(SLF_SC_XFACE=$xf timeout 300 python /tmp/scs4.py $n $pat 2>&1 | grep 'Total MLUPS\|Error\|error' | tail -1)" | tee -a $O/scs_x_slabs.txt
        done
      done
    done
    ;;
  r6s5)   # the planes with node maps (walls): all Shan-Chen tests + the processes
    ( time timeout 1500 python -m pytest tests/test_gpu_sc.py -m gpu -q --durations=5 ) > $O/pytest_sc_all.log 2>&1; tail -15 $O/pytest_sc_all.log
    ( time timeout 900 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q -k "shan_chen" --durations=5 ) > $O/pytest_sc_ranks.log 2>&1; tail -8 $O/pytest_sc_ranks.log
    ;;
  r6s6)   # the x-slab configurations of bench_configs + the Shan-Chen tests with the last host-side changes
    timeout 900 python tools/bench_configs.py --only 4x4,8x4,5x3,4,8a,5a 2>/dev/null | grep '^{' > $O/configs_x_slabs.jsonl; cut -c1-220 $O/configs_x_slabs.jsonl
    ( time timeout 1500 python -m pytest tests/test_gpu_sc.py tests/test_gpu_two_ranks.py -m gpu -q -k "sc or shan" --durations=5 ) > $O/pytest_sc_last.log 2>&1; tail -6 $O/pytest_sc_last.log
    ;;
  r6s7)   # slot sweep per boundary-condition level, f64 launch bounds, LES kernels with 512-thread workgroups
    ( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py tests/test_gpu_reg_les.py tests/test_gpu_sc.py -m gpu -q -k "indirect or reg_les or turb or regularized or smagorinsky or roundoff or les" --durations=5 ) > $O/pytest_indirect_les.log 2>&1; tail -8 $O/pytest_indirect_les.log
    timeout 900 python tools/bench_configs.py --only 7a,7c 2>/dev/null | grep '^{' > $O/configs_indirect.jsonl; cut -c1-200 $O/configs_indirect.jsonl
    for opt in "" "--subgrid=les-smagorinsky" "--regularized"; do
      echo "ldc_3d 256^3 AA $opt: $(timeout 300 python examples/ldc_3d.py --lat_nx=256 --lat_ny=256 --lat_nz=256 --visc=0.01 --access_pattern=AA --mode=benchmark --max_iters=600 --benchmark_sample_from=200 --perf_stats_every=0 $opt 2>&1 | grep 'Total MLUPS')" | tee -a $O/les_rates.txt
    done
    ;;
  r6end)   # after the last kernel-source change of the round: PMC passes, bench lines, kernel statistics, every GPU test
    for pat in AA AB; do
      PMC_SIZES_ONLY=1 BENCH_ARGS="--access_pattern $pat" bash tools/gpu.sh pmc; cp $O/pmc_summary.txt $O/pmc_sizes_${pat}_512_final.txt
      [ $pat = AA ] && kern="slf::fast_" || kern="slf::fast_row_kernel"
      python tools/traffic_update.py --from-pmc $O/pmc --kernel "$kern" --key D3Q19_bgk_f32_${pat}_512_fused
      rm -rf $O/pmc
    done
    cp profiles/traffic.json $O/traffic.json
    timeout 900 python bench.py 2>&1 | tail -1 > $O/bench_final.json; cut -c1-400 $O/bench_final.json
    timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $O/bench_driver_cmd_final.json; cut -c1-300 $O/bench_driver_cmd_final.json
    for pat in AA AB; do
      BENCH_ARGS="--access_pattern $pat" bash tools/gpu.sh trace; cp $O/kernel_stats.csv $O/kernel_stats_${pat}_final.csv; rm -rf $O/trace
    done
    ( time timeout 2700 python -m pytest tests -m gpu -q --durations=12 ) > $O/pytest_gpu_final.log 2>&1; tail -22 $O/pytest_gpu_final.log
    ;;
  r6s8)   # bytes and kernel times of the Shan-Chen x-slab configurations (planes) beside their undivided twins
    TRACE_CONFIGS="4x4 8x4" bash tools/gpu.sh tracecfg pmccfg; rm -rf $O/trace_cfg*/ $O/pmc_cfg*/
    ;;
  r6s9)   # ghost-pull skip of the Shan-Chen planes in the odd in-place step: parity, bytes, rates
    ( time timeout 1500 python -m pytest tests/test_gpu_sc.py tests/test_gpu_two_ranks.py -m gpu -q -k "sc or shan" --durations=3 ) > $O/pytest_sc_skip.log 2>&1; tail -6 $O/pytest_sc_skip.log
    TRACE_CONFIGS="8x4" bash tools/gpu.sh pmccfg; rm -rf $O/pmc_cfg*/
    cat > /tmp/sc4.py <<PYEOF
import sys
sys.path.insert(0, '$GRAFT_REPO_ROOT')
from examples.binary_fluid.sc_separation_3d import SeparationSim
from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
n = int(sys.argv[1])
c = LBSimulationController(SeparationSim, EqualSubdomainsGeometry3D, default_config=dict(lat_nx=256, lat_ny=256, lat_nz=256, subdomains=n, conn_axis='x', access_pattern=sys.argv[2], mode='benchmark', max_iters=300, benchmark_sample_from=100, perf_stats_every=0))
c.run(ignore_cmdline=True)
PYEOF
    for rep in 1 2; do
      echo "binary AA planes 4 x-slabs: $(timeout 300 python /tmp/sc4.py 4 AA 2>&1 | grep 'Total MLUPS')" | tee -a $O/sc_skip_rates.txt
    done
    timeout 900 python tools/bench_configs.py --only 8x4 2>/dev/null | grep '^{' | cut -c1-160 | tee -a $O/sc_skip_rates.txt
    ;;
  r6s10)  # the pipe in 3 x-slabs with cuts on whole waves (192 / 128 / 192) against 160 / 192 / 160
    P="--lat_nx=512 --lat_ny=256 --lat_nz=256 --subdomains=3 --conn_axis=x --visc=0.05 --access_pattern=AA --mode=benchmark --max_iters=600 --benchmark_sample_from=200 --perf_stats_every=0"
    for rep in 1 2; do
      echo "pipe AA align 32 tol 0.15: $(timeout 300 python examples/poiseuille_3d.py $P 2>&1 | grep 'Total MLUPS')" | tee -a $O/slab_align64.txt
      echo "pipe AA align 64 tol 0.3: $(timeout 300 python examples/poiseuille_3d.py $P --slab_align=64 --slab_tolerance=0.3 2>&1 | grep 'Total MLUPS')" | tee -a $O/slab_align64.txt
    done
    ;;
  r6s11)  # x cuts on whole waves for slabs of one process on one device (the new default): every multi-subdomain GPU test
    ( time timeout 2400 python -m pytest tests -m gpu -q -k "subdomain or slab or group or planes or ranks or xface or config4 or example or launch" --durations=5 ) > $O/pytest_slabs_waves.log 2>&1; tail -8 $O/pytest_slabs_waves.log
    timeout 900 python tools/bench_configs.py --only 5x3,3g8 2>/dev/null | grep '^{' | cut -c1-200 | tee $O/configs_slab_waves.txt
    ;;
  r6all)  # every GPU test + smoke, as the driver runs them at round end
    bash tools/gpu.sh smoke
    ( time timeout 2700 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu_final.log 2>&1; tail -16 $O/pytest_gpu_final.log
    ;;
  r6s12)  # binary Shan-Chen 256^3, one PROCESS per x-slab on the one GPU (planes in the neighbour's memory, peer transport)
    A="--lat_nx=256 --lat_ny=256 --lat_nz=256 --conn_axis=x --mode=benchmark --max_iters=400 --benchmark_sample_from=150 --perf_stats_every=0"
    for n in 4 2; do
      g=$(python -c "print(' '.join(['0'] * $n))")
      for xf in 1 0; do
        echo "processes $n planes $xf: $(SLF_SC_XFACE=$xf timeout 600 python tests/_sc_ranks_script.py $A --subdomains=$n --gpus $g 2>&1 | grep 'Total MLUPS\|rror' | tail -2)" | tee -a $O/sc_processes.txt
      done
    done
    echo "one process 4 slabs: $(timeout 600 python tests/_sc_ranks_script.py $A --subdomains=4 --gpus 0 2>&1 | grep 'Total MLUPS')" | tee -a $O/sc_processes.txt
    echo "undivided: $(timeout 600 python tests/_sc_ranks_script.py $A --subdomains=1 --gpus 0 2>&1 | grep 'Total MLUPS')" | tee -a $O/sc_processes.txt
    ;;
  r6s13)  # closed box in three x-slabs through the planes
    ( time timeout 900 python -m pytest tests/test_gpu_sc.py -m gpu -q -k "closed_box" --durations=3 ) > $O/pytest_sc_box.log 2>&1; tail -12 $O/pytest_sc_box.log
    ;;
  r6final)   # round-6 evidence visit: everything DESIGN.md / profiles/traffic.json quote for the shipped kernels
    export SLF_PEER_TIMEOUT_S=60
    bash tools/gpu.sh host smoke
    for pat in AA AB; do
      PMC_SIZES_ONLY=1 BENCH_ARGS="--access_pattern $pat" bash tools/gpu.sh pmc; cp $O/pmc_summary.txt $O/pmc_sizes_${pat}_512_final.txt
      [ $pat = AA ] && kern="slf::fast_" || kern="slf::fast_row_kernel"
      python tools/traffic_update.py --from-pmc $O/pmc --kernel "$kern" --key D3Q19_bgk_f32_${pat}_512_fused
      rm -rf $O/pmc
    done
    cp profiles/traffic.json $O/traffic.json
    timeout 900 python bench.py 2>&1 | tail -1 > $O/bench_final.json; cut -c1-400 $O/bench_final.json
    timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $O/bench_driver_cmd_final.json; cut -c1-300 $O/bench_driver_cmd_final.json
    for pat in AA AB; do
      BENCH_ARGS="--access_pattern $pat" bash tools/gpu.sh trace; cp $O/kernel_stats.csv $O/kernel_stats_${pat}_final.csv; rm -rf $O/trace
    done
    # the same box, back to back: the 256^3 box of BASELINE config 2 through bench.py and through the host stack, and the headline again
    timeout 600 python bench.py --size 256 --no_cpu_baseline --no_gpu_state 2>&1 | tail -1 > $O/bench_256.json; cut -c1-200 $O/bench_256.json
    timeout 1500 python tools/bench_configs.py 2>/dev/null | grep '^{' > $O/configs_final.jsonl; cut -c1-150 $O/configs_final.jsonl
    timeout 900 python tools/bench_configs.py --only 6xa,6xb,7a,7b,7c,7d,8a,8b,9a,9b 2>/dev/null | grep '^{' >> $O/configs_final.jsonl; tail -10 $O/configs_final.jsonl | cut -c1-150
    timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>&1 | tail -1 > $O/bench_driver_cmd_after_configs.json; cut -c1-200 $O/bench_driver_cmd_after_configs.json
    bash tools/gpu.sh torchrun; mv $O/torchrun.jsonl $O/torchrun_final_peer.jsonl
    SLF_HALO_TRANSPORT=rccl bash tools/gpu.sh torchrun; mv $O/torchrun.jsonl $O/torchrun_final_rccl.jsonl
    ( time timeout 2700 python -m pytest tests -m gpu -q --durations=12 ) > $O/pytest_gpu_final.log 2>&1; tail -22 $O/pytest_gpu_final.log
    ;;
  *) echo "unknown visit $NAME" ;;
esac
