"""Where the device's time goes in a rocprofv3 --kernel-trace CSV: per kernel the launches, mean and total time; how much of
the traced span the device was executing ANY kernel (union of the intervals), and the idle gaps between them -- the
launch-bound part of a run that steps several subdomains from one process.

    python tools/probe/trace_busy.py <dir with *kernel_trace.csv> [fraction of the span to skip at the start, default 0.4]
"""
import csv
import glob
import os
import sys

import numpy as np


def main():
    d = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    if not rows:
        raise SystemExit('no kernel_trace.csv below %s' % d)
    rows.sort()
    t_first, t_last = rows[0][0], max(e for _, e, _ in rows)
    t0 = t_first + skip * (t_last - t_first)
    rows = [r for r in rows if r[0] >= t0]
    span = (max(e for _, e, _ in rows) - rows[0][0]) / 1e3
    names = {}
    for s, e, n in rows:
        n = n.replace('void slf::', '').replace('slf::', '')
        names.setdefault(n[:96], []).append((e - s) / 1e3)
    print('span %.1f ms, %d launches (the first %.0f %% of the trace skipped)' % (span / 1e3, len(rows), skip * 100))
    print('%-96s %7s %9s %9s %6s' % ('kernel', 'count', 'mean_us', 'total_ms', 'share'))
    for n, ds in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:16]:
        print('%-96s %7d %9.2f %9.2f %6.3f' % (n, len(ds), np.mean(ds), sum(ds) / 1e3, sum(ds) / span))
    # union of the busy intervals
    busy, gaps = 0.0, []
    cs, ce = rows[0][0], rows[0][1]
    for s, e, _ in rows[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append((s - ce) / 1e3)
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    gaps = np.array(gaps) if gaps else np.zeros(1)
    print('device busy %.4f of the span; %d idle gaps: total %.2f ms, mean %.2f us, median %.2f, p90 %.2f, max %.1f'
          % (busy / 1e3 / span, len(gaps), gaps.sum() / 1e3, gaps.mean(), np.median(gaps), np.percentile(gaps, 90), gaps.max()))
    small = sum(sum(ds) for n, ds in names.items() if np.mean(ds) < 20.0)
    print('kernels shorter than 20 us on average: %.3f of the span in total' % (small / span))


if __name__ == '__main__':
    main()
