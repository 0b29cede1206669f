"""The bench.py output contract (one JSON line), checked on the line committed from the last GPU run of the
round, and the argument handling that can be checked without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_every_contract_field():
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r06', 'bench_driver_cmd_final.json')))    # the driver's command line
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'MLUPS' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic'
    assert d['vs_baseline'] is None and d['dtype'] == 'f32' and 'workload' in d['config'] and 'model' not in d['config']
    assert 'D3Q19 BGK 512^3' in d['metric']
    c = d['config']
    # round 4: blocks of exactly K steps, repeated until at least half a second has been timed;
    # round 5: the headline is the MEDIAN block, the fastest one an extra key
    assert c['value_is'].startswith('median block of exactly K steps') and c['timed_blocks'] >= c['repeats']
    assert c['timed_seconds'] >= 0.5 and abs(c['timed_seconds'] / c['timed_blocks'] - d['ms_per_step'] * d['steps'] * 1e-3) < 0.2 * c['timed_seconds'] / c['timed_blocks']
    assert min(c['runs_mlups']) <= c['median_mlups'] <= max(c['runs_mlups']) and abs(max(c['runs_mlups']) - d['best_value']) < 1.0
    # round 6: `value_kind` says in words what `value` is (the duplicate key median_value is gone)
    assert d['value'] == c['median_mlups'] and d['value'] <= d['best_value'] < 1.01 * d['value'] and 'median_value' not in d
    assert d['value_kind'].startswith('median block')
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    # achieved = algorithmic bytes per launch / kernel time; value = updates / wall time
    assert abs(r['achieved'] - 512 ** 3 * r['bytes_per_update'] / (r['kernel_ms'] * 1e-3) / 1e9) < 1.0
    assert abs(d['value'] - 512 ** 3 / (d['ms_per_step'] * 1e-3) * 1e-6) / d['value'] < 1e-3
    assert r['kernel_ms'] <= d['ms_per_step'] * 1.001
    assert r['traffic'] is None or 0.9 < r['traffic'] / (512 ** 3 * 152) < 1.2
    # round 3: the line validates itself and carries the runner-path leg
    assert c['validated'] is True and all(v['populations_bit_identical'] and v['ok'] for v in c['validation'].values())
    assert all(v['populations_compared'] > 2e7 and v['mass_rel_drift'] < 1e-5 for v in c['validation'].values())
    rp = c['runner_path']
    assert 'SubdomainRunner.step()' in rp['through'] and abs(rp['vs_value'] - 1.0) < 0.03, rp
    b = d['cpu_baseline']
    assert b['kind'] == 'port' and b['unit'] == 'MLUPS' and b['cores'] >= 1 and 'oracle/lbm_fast.c' in b['sample']
    assert b['single_thread_mlups'] > 0 and b['config1_d2q9_256x256_mlups'] > 0 and 0 < b['numpy_twin_mlups'] < b['value']


def test_committed_traffic_belongs_to_the_committed_kernels():
    """profiles/traffic.json is stamped with the hash of the kernel sources in this tree: a kernel change without new PMC
    passes makes bench.py report `traffic: null` -- and this test say so before a round ends."""
    from sailfish_amd import build as slf_build
    t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    assert t['_csrc_sha256'] == slf_build.source_hash(), 'kernel sources changed since the PMC passes: repeat them (tools/gpu.sh pmc, tools/traffic_update.py)'
    for k in ('D3Q19_bgk_f32_AB_512_fused', 'D3Q19_bgk_f32_AA_512_fused'):
        assert 0.99 < t[k] / (512 ** 3 * 152) < 1.05


def test_strong_scaling_arguments_are_checked_before_any_gpu_work():
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    src = open(os.path.join(ROOT, 'bench.py')).read()
    for flag in ('--scaling', '--domain', '--axis', '--force_distributed'):
        assert flag in src
    from sailfish_amd.slab import SlabPlan  # noqa: F401  (the slab driver imports without a GPU)


def test_bench_launches_its_own_ranks_and_refuses_to_run_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (torch.distributed.run); here,
    without a GPU, every rank refuses -- there is no CPU fallback -- and the status is passed on."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=900)
    assert res.returncode != 0 and b'bench.py needs a GPU' in res.stdout and b'torch.distributed' in res.stdout, res.stdout[-2000:]


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_value_is_the_median_block_not_the_best():
    """VERDICT r4 / ADVICE r4: the headline is the median timed block; the fastest one is an extra key."""
    bench = _bench_module()
    blocks = [0.0660, 0.0644, 0.0651, 0.0649, 0.0700]
    assert blocks[bench.median_block(blocks)] == 0.0651
    assert blocks[bench.median_block(blocks[:4])] == 0.0649          # even count: the lower middle one, a block that ran
    assert bench.median_block([0.07]) == 0
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert "'best_value'" in src and "'value': round(to_mlups(elapsed), 1)" in src and 'np.argmin(blocks)' not in src


def test_traffic_is_reported_only_for_the_sources_it_was_measured_with(tmp_path):
    """profiles/traffic.json carries the sha256 of sailfish_amd/csrc/*; any other state of the sources -> null."""
    from sailfish_amd import build as slf_build
    bench = _bench_module()
    good = tmp_path / 'good.json'
    good.write_text(json.dumps({'_csrc_sha256': slf_build.source_hash(), 'K': 123}))
    stale = tmp_path / 'stale.json'
    stale.write_text(json.dumps({'_csrc_sha256': '0' * 64, 'K': 123}))
    unstamped = tmp_path / 'unstamped.json'
    unstamped.write_text(json.dumps({'K': 123}))
    assert bench.load_traffic('K', str(good)) == 123 and bench.load_traffic('other', str(good)) is None
    assert bench.load_traffic('K', str(stale)) is None and bench.load_traffic('K', str(unstamped)) is None
    assert bench.load_traffic('K', str(tmp_path / 'missing.json')) is None
    h = slf_build.source_hash()
    assert len(h) == 64 and h == slf_build.source_hash()


def test_rccl_ranks_come_from_the_communicator_not_from_the_process_group():
    """`rccl_ranks` = ncclCommCount of the communicator the halos travel over (C ABI slf_comm_count); a gloo group
    (several ranks on one GPU, host staging) reports 0 whatever its world size."""
    bench = _bench_module()

    class Rccl(object):
        def count(self):
            return (8, 3)

    class Direct(object):
        direct = True
        rccl = Rccl()

    class Staged(object):
        pass

    assert bench.rccl_ranks_of(Direct(), 'nccl', 8) == 8
    assert bench.rccl_ranks_of(Staged(), 'gloo', 4) == 0 and bench.rccl_ranks_of(None, None, 1) == 0
    assert bench.rccl_ranks_of(Staged(), 'nccl', 2) == 2            # SLF_HALO_TRANSPORT=torch: torch's own communicator
    hdr = open(os.path.join(ROOT, 'include', 'sailfish_hip.h')).read()
    assert 'int slf_comm_count(slf_comm* comm, int* nranks, int* rank);' in hdr
