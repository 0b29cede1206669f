#!/usr/bin/env python3
"""profiles/traffic.json <- the HBM bytes per sweep launch of a rocprofv3 --pmc run (tools/pmc_summary.py's figures),
stamped with the sha256 of the kernel sources they were measured with (sailfish_amd/build.py source_hash()).  bench.py
reports `roofline.traffic` only while that stamp matches the sources of the library it runs.

    python tools/traffic_update.py KEY=BYTES [KEY=BYTES ...] [--comment TEXT]
    python tools/traffic_update.py --from-pmc DIR --kernel SUBSTRING --key KEY      # read + write bytes of one kernel

KEY is bench.py's workload key, e.g. D3Q19_bgk_f32_AB_512_fused.  Every entry of the file belongs to ONE state of the
sources: entries that are not given again are dropped when the stamp changes.
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pmc_bytes(root, kernel):
    """(read, write) bytes per launch of the kernels whose name contains `kernel`: FETCH_SIZE KiB x 1024 x 2 (gfx950
    correction, /opt/skills/guides/MI355X_MICROARCH.md) and WRITE_SIZE KiB x 1024, averaged over the launches."""
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
        for row in csv.DictReader(open(f)):
            if kernel in row['Kernel_Name'] and row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                a = agg[row['Counter_Name']]
                a[0] += float(row['Counter_Value'])
                a[1] += 1
    if not agg['FETCH_SIZE'][1] or not agg['WRITE_SIZE'][1]:
        raise SystemExit('no FETCH_SIZE / WRITE_SIZE rows for a kernel matching %r below %s' % (kernel, root))
    rd = agg['FETCH_SIZE'][0] / agg['FETCH_SIZE'][1] * 1024 * 2
    wr = agg['WRITE_SIZE'][0] / agg['WRITE_SIZE'][1] * 1024
    return rd, wr


def main():
    from sailfish_amd import build as slf_build
    ap = argparse.ArgumentParser()
    ap.add_argument('pairs', nargs='*')
    ap.add_argument('--from-pmc')
    ap.add_argument('--kernel')
    ap.add_argument('--key')
    ap.add_argument('--comment')
    ap.add_argument('--file', default=os.path.join(ROOT, 'profiles', 'traffic.json'))
    a = ap.parse_args()
    stamp = slf_build.source_hash()
    try:
        data = json.load(open(a.file))
    except Exception:  # noqa: BLE001
        data = {}
    if data.get('_csrc_sha256') != stamp:
        data = dict((k, v) for k, v in data.items() if k == '_comment')
    data['_csrc_sha256'] = stamp
    for pair in a.pairs:
        k, v = pair.split('=')
        data[k] = int(float(v))
    if a.from_pmc:
        rd, wr = pmc_bytes(a.from_pmc, a.kernel or '')
        data[a.key] = int(rd + wr)
        print('%s: read %.4f GB + written %.4f GB per launch' % (a.key, rd / 1e9, wr / 1e9))
    if a.comment:
        data['_comment'] = a.comment
    with open(a.file, 'w') as fh:
        json.dump(data, fh, indent=1)
        fh.write('\n')
    print('stamped', stamp[:16], sorted(k for k in data if not k.startswith('_')))


if __name__ == '__main__':
    main()
