#!/usr/bin/env python3
"""Box probe (GPU box only): why does the same sweep kernel take 3.35 ms on one visit and 3.8 ms on another?

In ONE process, on the same allocations, records
  * a per-launch time series (HIP events around every launch) of the three headline sweep kernels (even AA, odd AA,
    AB) and of plain 16-byte copy / read / write kernels over the same number of bytes (tools/probe/probe_kernels.hip);
  * GPU clocks, power and temperatures sampled from sysfs (hwmon + pp_dpm_*) in a background thread at ~50 Hz,
    time-stamped on the same host clock as the batch boundaries of the time series;
  * `amd-smi metric` / `rocm-smi` dumps before and after;
  * optionally the same again under a pinned performance level (rocm-smi --setperflevel high /
    --setperfdeterminism), restoring `auto` afterwards.

Output: one JSON file (summary statistics + the raw series) and a readable text summary.

    python tools/probe/box_probe.py --size 512 --launches 400 --out gpurun_out/box_probe
"""
import argparse
import ctypes
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)


def build_probe_lib():
    src = os.path.join(ROOT, 'tools', 'probe', 'probe_kernels.hip')
    out = os.path.join(ROOT, 'tools', 'probe', 'libprobe.so')
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-o', out, src])
    return out


class Sampler(threading.Thread):
    """Samples every readable hwmon / dpm file of the first amdgpu card."""

    def __init__(self, period=0.02):
        threading.Thread.__init__(self, daemon=True)
        self.period = period
        self.samples = []
        self.stop_flag = False
        self.files = {}
        for dev in sorted(glob.glob('/sys/class/drm/card*/device')):
            if not os.path.exists(os.path.join(dev, 'pp_dpm_sclk')):
                continue
            for hw in glob.glob(os.path.join(dev, 'hwmon', 'hwmon*')):
                for pat in ('freq*_input', 'power*_average', 'power*_input', 'temp*_input', 'in*_input'):
                    for f in sorted(glob.glob(os.path.join(hw, pat))):
                        label = os.path.basename(f)
                        lf = f.replace('_input', '_label').replace('_average', '_label')
                        try:
                            label = os.path.basename(f).split('_')[0] + ':' + open(lf).read().strip()
                        except OSError:
                            pass
                        self.files[label] = f
            for name in ('gpu_busy_percent', 'mem_busy_percent'):
                f = os.path.join(dev, name)
                if os.path.exists(f):
                    self.files[name] = f
            self.dpm = dict((n, os.path.join(dev, n)) for n in ('pp_dpm_sclk', 'pp_dpm_mclk', 'pp_dpm_fclk', 'pp_dpm_socclk')
                            if os.path.exists(os.path.join(dev, n)))
            self.dev = dev
            break
        else:
            self.dpm, self.dev = {}, None

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return fh.read().strip()
        except OSError:
            return None

    def dpm_state(self):
        out = {}
        for n, f in self.dpm.items():
            txt = self._read(f) or ''
            out[n] = [ln.strip() for ln in txt.splitlines()]
        return out

    def run(self):
        while not self.stop_flag:
            row = {'t': time.perf_counter()}
            for label, f in self.files.items():
                v = self._read(f)
                try:
                    row[label] = int(v)
                except (TypeError, ValueError):
                    pass
            self.samples.append(row)
            time.sleep(self.period)

    def window(self, t0, t1):
        rows = [r for r in self.samples if t0 <= r['t'] <= t1]
        out = {}
        for key in (rows[0].keys() if rows else []):
            if key == 't':
                continue
            vals = np.array([r[key] for r in rows if key in r], dtype=np.float64)
            if len(vals):
                out[key] = {'min': float(vals.min()), 'mean': float(vals.mean()), 'max': float(vals.max())}
        return out


def sh(cmd, timeout=30):
    try:
        p = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
        return p.stdout.decode(errors='replace')
    except Exception as e:  # noqa: BLE001
        return 'FAILED: %s' % e


def stats(ms):
    a = np.asarray(ms, dtype=np.float64)
    return {'n': int(len(a)), 'min': float(a.min()), 'p10': float(np.percentile(a, 10)), 'median': float(np.median(a)),
            'p90': float(np.percentile(a, 90)), 'max': float(a.max()), 'mean': float(a.mean()), 'std': float(a.std())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--launches', type=int, default=400)
    ap.add_argument('--batch', type=int, default=20, help='launches between two host-clock marks')
    ap.add_argument('--out', default='gpurun_out/box_probe')
    ap.add_argument('--pin', default='', help='comma list of extra passes: high (setperflevel high), '
                                                'det:<MHz> (setperfdeterminism), low')
    ap.add_argument('--skip_smi', action='store_true')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    from sailfish_amd import hipabi, sym
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.box import make_box_desc

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    probe = ctypes.CDLL(build_probe_lib())
    probe.probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                   ctypes.c_void_p]
    n = args.size
    grid = sym.D3Q19
    size = (n, n, n)
    desc_aa = make_box_desc(grid, size, precision='single', access_pattern='AA', visc=1.0 / 6.0, periodic_fused=[1, 1, 1])
    desc_ab = make_box_desc(grid, size, precision='single', access_pattern='AB', visc=1.0 / 6.0, periodic_fused=[1, 1, 1])
    nodes = desc_aa.arr_nx * desc_aa.arr_ny * desc_aa.arr_nz
    off = b.dist_align_offset(4)
    dist_a = b.alloc_buf(size=19 * nodes * 4, align_offset=off)
    dist_b = b.alloc_buf(size=19 * nodes * 4, align_offset=off)
    shape = (desc_aa.arr_nz, desc_aa.arr_ny, desc_aa.arr_nx)
    rho = (1.0 + 1e-3 * np.random.RandomState(1).rand(*shape)).astype(np.float32)
    v = np.zeros(shape, dtype=np.float32)
    g_rho = b.alloc_buf(like=rho)
    g_v = [b.alloc_buf(like=v) for _ in range(3)]
    stream = b.make_stream()
    mod_aa, mod_ab = b.build(desc_aa), b.build(desc_ab)
    for d in (dist_a, dist_b):
        k = b.get_kernel(mod_aa, 'SetInitialConditions', (64,), [d] + g_v + [g_rho, 0], 'PPPPPP')
        b.run_kernel(k, None, stream)
    stream.synchronize()
    sig = 'PPPPPPPi'
    k_aa = b.get_kernel(mod_aa, 'CollideAndPropagate', (64,), [0, dist_a, dist_a, g_rho] + g_v + [0], sig, needs_iteration=True)
    k_ab = [b.get_kernel(mod_ab, 'CollideAndPropagate', (64,), [0, dist_a, dist_b, g_rho] + g_v + [0], sig),
            b.get_kernel(mod_ab, 'CollideAndPropagate', (64,), [0, dist_b, dist_a, g_rho] + g_v + [0], sig)]
    sweep_bytes = n ** 3 * 152
    copy_bytes = (n ** 3 * 76) // 4096 * 4096      # read N + write N  == the sweep's 152 B per node
    native = ctypes.c_void_p(stream.native)

    def launcher(name):
        if name == 'aa_even':
            def f(i):
                b._lib.slf_kernel_set_iteration(k_aa.handle, 0)
                b.run_kernel(k_aa, None, stream)
        elif name == 'aa_odd':
            def f(i):
                b._lib.slf_kernel_set_iteration(k_aa.handle, 1)
                b.run_kernel(k_aa, None, stream)
        elif name == 'aa':
            def f(i):
                b._lib.slf_kernel_set_iteration(k_aa.handle, i)
                b.run_kernel(k_aa, None, stream)
        elif name == 'ab':
            def f(i):
                b.run_kernel(k_ab[i & 1], None, stream)
        else:
            kind, nt = {'copy': (0, 0), 'copy_nt': (0, 3), 'read': (1, 0), 'read_nt': (1, 1), 'write': (2, 0),
                        'write_nt': (2, 2)}[name]

            def f(i):
                rc = probe.probe_launch(kind, nt, ctypes.c_void_p(dist_b), ctypes.c_void_p(dist_a), copy_bytes, native)
                assert rc == 0, rc
        return f

    def bytes_of(name):
        if name in ('aa_even', 'aa_odd', 'aa', 'ab'):
            return sweep_bytes
        return copy_bytes * (2 if name.startswith('copy') else 1)

    sampler = Sampler()
    sampler.start()
    report = {'size': n, 'launches': args.launches, 'device': b.info, 'sysfs_files': sorted(sampler.files),
              'dpm_before': sampler.dpm_state(), 'passes': []}
    text = []

    def say(s):
        print(s, flush=True)
        text.append(s)

    say('device: %s' % b.info)
    say('sysfs sensors: %s' % ', '.join(sorted(sampler.files)))
    if not args.skip_smi:
        report['amd_smi_before'] = sh('amd-smi metric 2>&1 | head -150')
        report['amd_smi_static'] = sh('amd-smi static --limit --clock --vram 2>&1 | head -120')
        report['rocm_smi_before'] = sh('rocm-smi --showclocks --showpower --showtemp --showperflevel --showmemuse 2>&1 | head -80')

    def one_pass(tag, names):
        res = {'tag': tag, 'dpm': sampler.dpm_state(), 'kernels': {}}
        say('--- pass %s' % tag)
        for name in names:
            f = launcher(name)
            for i in range(6):
                f(i)
            stream.synchronize()
            evs = [b.make_event(stream, timing=True)]
            marks = [time.perf_counter()]
            for i in range(args.launches):
                f(i)
                evs.append(b.make_event(stream, timing=True))
                if (i + 1) % args.batch == 0:
                    evs[-1].synchronize()
                    marks.append(time.perf_counter())
            evs[-1].synchronize()
            t_end = time.perf_counter()
            ms = [evs[j + 1].time_since(evs[j]) for j in range(args.launches)]
            st = stats(ms)
            gbs = bytes_of(name) / st['median'] / 1e6
            sens = sampler.window(marks[0], t_end)
            res['kernels'][name] = {'stats_ms': st, 'median_GBs': gbs, 'best_GBs': bytes_of(name) / st['min'] / 1e6,
                                    'series_ms': [round(x, 4) for x in ms], 'marks': marks, 'sensors': sens}
            key = lambda k: sens.get(k, {}).get('mean', float('nan'))  # noqa: E731
            sclk = next((key(k) for k in sens if k.startswith('freq1')), float('nan'))
            mclk = next((key(k) for k in sens if k.startswith('freq2')), float('nan'))
            pw = next((key(k) for k in sens if k.startswith('power1')), float('nan'))
            say('%-9s median %.3f ms (%.0f GB/s)  min %.3f  p90 %.3f  max %.3f  std %.3f | sclk %.0f MHz mclk %.0f MHz power %.0f W'
                % (name, st['median'], gbs, st['min'], st['p90'], st['max'], st['std'], sclk / 1e6, mclk / 1e6, pw / 1e6))
        report['passes'].append(res)

    names = ['copy_nt', 'copy', 'read_nt', 'write_nt', 'aa_even', 'aa_odd', 'aa', 'ab', 'copy_nt']
    one_pass('auto', names)
    for pin in [p for p in args.pin.split(',') if p]:
        if pin == 'high':
            out = sh('rocm-smi --setperflevel high 2>&1 | tail -5')
        elif pin == 'low':
            out = sh('rocm-smi --setperflevel low 2>&1 | tail -5')
        elif pin.startswith('det:'):
            out = sh('rocm-smi --setperfdeterminism %s 2>&1 | tail -5' % pin[4:])
        else:
            continue
        say('pin %s: %s' % (pin, out.strip().replace('\n', ' | ')))
        time.sleep(0.5)
        one_pass(pin, ['copy_nt', 'aa_even', 'aa_odd', 'ab'])
        say('restore: %s' % sh('rocm-smi --resetperfdeterminism 2>&1 | tail -2; rocm-smi --setperflevel auto 2>&1 | tail -2').strip().replace('\n', ' | '))
    if args.pin:
        one_pass('auto_again', ['copy_nt', 'aa_even', 'aa_odd', 'ab'])
    if not args.skip_smi:
        report['amd_smi_after'] = sh('amd-smi metric 2>&1 | head -150')
    sampler.stop_flag = True
    report['dpm_after'] = sampler.dpm_state()
    with open(os.path.join(args.out, 'box_probe.json'), 'w') as fh:
        json.dump(report, fh)
    with open(os.path.join(args.out, 'box_probe.txt'), 'w') as fh:
        fh.write('\n'.join(text) + '\n')
        for k in ('amd_smi_static', 'amd_smi_before', 'rocm_smi_before', 'amd_smi_after'):
            if k in report:
                fh.write('\n===== %s =====\n%s\n' % (k, report[k]))
        fh.write('\n===== dpm before =====\n%s\n' % json.dumps(report['dpm_before'], indent=1))


if __name__ == '__main__':
    main()
