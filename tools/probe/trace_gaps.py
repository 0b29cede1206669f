"""Timeline summary of a rocprofv3 --kernel-trace CSV of the chunked x-slab step: per kernel name the count and mean
duration, and for the sweep kernels the gaps between consecutive launches (end of one -> start of the next), split
into gaps inside a step and between steps; how much of the communication kernels' run time overlaps a sweep kernel.

    python tools/probe/trace_gaps.py <dir with *kernel_trace.csv> [skip_first_n_sweeps]
"""
import csv
import glob
import os
import sys

import numpy as np


def main():
    d = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''), r.get('Stream_Id', '')))
    rows.sort()
    names = {}
    for s, e, n, q, st in rows:
        names.setdefault(n[:70], []).append(e - s)
    print('kernel                                                                     count   mean_us')
    for n, ds in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print('%-72s %7d %9.2f' % (n, len(ds), np.mean(ds) / 1e3))
    is_sweep = lambda n: 'row_kernel' in n or 'even_kernel' in n or 'sweep_kernel' in n  # noqa: E731
    sw = [(s, e) for s, e, n, q, st in rows if is_sweep(n)][skip:]
    comm = [(s, e) for s, e, n, q, st in rows if 'ccl' in n.lower()]
    if len(sw) < 10:
        return
    gaps = np.array([sw[i + 1][0] - sw[i][1] for i in range(len(sw) - 1)]) / 1e3
    durs = np.array([e - s for s, e in sw]) / 1e3
    t0, t1 = sw[0][0], sw[-1][1]
    print('sweep launches %d: mean duration %.2f us, busy %.4f of the span, gaps mean %.2f us median %.2f p90 %.2f max %.2f'
          % (len(sw), durs.mean(), durs.sum() * 1e3 / (t1 - t0), gaps.mean(), np.median(gaps), np.percentile(gaps, 90), gaps.max()))
    hist = np.histogram(gaps, bins=[-1e9, 0, 2, 5, 10, 20, 50, 100, 1e9])[0]
    print('gap histogram (<0, 0-2, 2-5, 5-10, 10-20, 20-50, 50-100, >100 us):', hist.tolist())
    # overlap of communication kernels with sweep kernels
    if comm:
        comm = [c for c in comm if c[0] >= t0 and c[1] <= t1]
        starts = np.array([s for s, e in sw])
        ends = np.array([e for s, e in sw])
        tot = ov = 0
        for s, e in comm:
            tot += e - s
            i0 = np.searchsorted(ends, s)
            for i in range(i0, len(sw)):
                if starts[i] >= e:
                    break
                ov += max(0, min(e, ends[i]) - max(s, starts[i]))
        print('comm kernels %d: mean %.2f us, %.3f of their run time overlaps a sweep kernel' % (len(comm), tot / max(1, len(comm)) / 1e3, ov / max(1, tot)))


if __name__ == '__main__':
    main()
