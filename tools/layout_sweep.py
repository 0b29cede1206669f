#!/usr/bin/env python3
"""Layout / scheduling knob sweep of the headline kernels at one size (GPU box): each configuration runs
tools/perf_probe.py in a fresh process (fresh allocations) and prints its line.

    python tools/layout_sweep.py --dims 512x512x512
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def run(dims, env, reps, modes, extra=()):
    e = dict(os.environ)
    e.update(env)
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'perf_probe.py'), '--dims', dims, '--variants', '11', '--blocks', '576',
           '--reps', str(reps), '--modes', modes] + list(extra)
    try:
        out = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300).stdout.decode()
    except subprocess.TimeoutExpired:
        out = 'TIMEOUT'
    line = [ln for ln in out.splitlines() if ln.startswith('variant')]
    tag = ' '.join('%s=%s' % kv for kv in sorted(env.items())) or 'default'
    print('%-44s %s' % (tag, line[-1] if line else out[-300:].replace('\n', ' | ')), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dims', default='512x512x512')
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--modes', default='even,odd,ab')
    ap.add_argument('--sets', default='base,order,lds,nxpad,nypad,distpad')
    args = ap.parse_args()
    sets = args.sets.split(',')
    cfgs = []
    if 'base' in sets:
        cfgs += [{}, {}]
    if 'order' in sets:
        cfgs += [{'SLF_ROW_ORDER': str(o)} for o in (1, 2, 3)]
    if 'lds' in sets:      # 160 KB LDS per CU: pad so that only k workgroups fit
        cfgs += [{'SLF_LDS_PAD': str(b)} for b in (40000, 52000, 64000, 80000)]
    if 'nxpad' in sets:
        cfgs += [{'SLF_ARR_NX_PAD': str(b)} for b in (32, 64, 96, 160)]
    if 'nypad' in sets:
        cfgs += [{'SLF_ARR_NY_PAD': str(b)} for b in (1, 2, 6, 14)]
    if 'distpad' in sets:
        cfgs += [{'SLF_DIST_PAD': str(b)} for b in (32, 1056, 4128, 65568, 524320)]
    if 'combo' in sets:
        cfgs += [{'SLF_ROW_ORDER': '1', 'SLF_LDS_PAD': '52000'}, {'SLF_ROW_ORDER': '3', 'SLF_LDS_PAD': '52000'},
                 {'SLF_ROW_ORDER': '1', 'SLF_ARR_NX_PAD': '32'}]
    for env in cfgs:
        run(args.dims, env, args.reps, args.modes)


if __name__ == '__main__':
    main()
