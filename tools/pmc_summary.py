#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files below a directory, plus the HBM bytes
per launch derived as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: read bytes = FETCH_SIZE
(KiB-units as rocprofv3 reports them) x 2, cross-checked with TCC_EA0_RDREQ x 128 B (32-B requests counted
separately); write bytes = WRITE_SIZE, cross-checked with TCC_EA0_WRREQ_64B x 64 B + the 32-B rest."""
import collections
import csv
import glob
import sys


def main(root):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(root + '/**/*counter_collection.csv', recursive=True)):
        for row in csv.DictReader(open(f)):
            k = (row['Kernel_Name'].split('(')[0][:70], row['Counter_Name'])
            agg[k][0] += float(row['Counter_Value'])
            agg[k][1] += 1
    kernels = sorted(set(k for k, _ in agg))
    for kn in kernels:
        c = dict((cn, v / n) for (k, cn), (v, n) in agg.items() if k == kn)
        n = max(nn for (k, cn), (v, nn) in agg.items() if k == kn)
        if n < 4:
            continue
        print('%s  (launches %d)' % (kn, n))
        for cn in sorted(c):
            print('    %-28s %.6g' % (cn, c[cn]))
        if 'FETCH_SIZE' in c:
            print('    read  bytes/launch (FETCH_SIZE KiB x 1024 x 2, gfx950 correction): %.4f GB' % (c['FETCH_SIZE'] * 1024 * 2 / 1e9))
        if 'TCC_EA0_RDREQ_sum' in c:
            r32 = c.get('TCC_EA0_RDREQ_32B_sum', 0.0)
            print('    read  bytes/launch (RDREQ: 32B x 32 + rest x 128):                  %.4f GB'
                  % (((c['TCC_EA0_RDREQ_sum'] - r32) * 128 + r32 * 32) / 1e9))
        if 'WRITE_SIZE' in c:
            print('    write bytes/launch (WRITE_SIZE KiB x 1024):                          %.4f GB' % (c['WRITE_SIZE'] * 1024 / 1e9))
        if 'TCC_EA0_WRREQ_sum' in c:
            w64 = c.get('TCC_EA0_WRREQ_64B_sum', 0.0)
            print('    write bytes/launch (WRREQ: 64B x 64 + rest x 32):                    %.4f GB'
                  % ((w64 * 64 + (c['TCC_EA0_WRREQ_sum'] - w64) * 32) / 1e9))
        if 'SQ_WAVES' in c and c['SQ_WAVES'] > 0:
            w = c['SQ_WAVES']
            print('    per wave: ' + ', '.join('%s %.0f' % (cn[9:], c[cn] / w) for cn in sorted(c) if cn.startswith('SQ_INSTS_')))
        if 'SQ_BUSY_CYCLES' in c and c['SQ_BUSY_CYCLES'] > 0:
            print('    of SQ_BUSY_CYCLES: ' + ', '.join('%s %.3f' % (cn[3:], c[cn] / c['SQ_BUSY_CYCLES']) for cn in sorted(c)
                                                      if cn.startswith(('SQ_ACTIVE_', 'SQ_WAIT', 'SQ_INST_CYCLES'))))
        if 'SQ_WAVE_CYCLES' in c and c['SQ_WAVE_CYCLES'] > 0:
            print('    of SQ_WAVE_CYCLES: ' + ', '.join('%s %.3f' % (cn[3:], c[cn] / c['SQ_WAVE_CYCLES']) for cn in sorted(c)
                                                      if cn.startswith(('SQ_ACTIVE_', 'SQ_WAIT'))))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '.')
