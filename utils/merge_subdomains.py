#!/usr/bin/env python3
"""Merges the per-subdomain output files of one iteration (or of all iterations) into a
single file covering the global domain (tool of the reference: utils/merge_subdomains.py).

    utils/merge_subdomains.py [--all] <base>.<subdomain>.<iter>.npz

Reads <base>.subdomains (written by the controller) for the position of every subdomain,
writes <base>.merged.<iter>.npz; nodes covered by no subdomain are NaN.
"""
import argparse
import glob
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from sailfish import io  # noqa: E402


def get_bounding_box(subdomains):
    """Global domain size, slowest axis first."""
    dim = subdomains[0].dim
    return tuple(max(s.end_location[a] for s in subdomains) for a in reversed(range(dim)))


def merge_subdomains(base, digits, it, save=True):
    with open(io.subdomains_filename(base), 'rb') as f:
        subdomains = pickle.load(f)
    box = get_bounding_box(subdomains)
    dim = len(box)
    out = {}
    for s in subdomains:
        data = np.load(io.filename(base, digits, s.id, it))
        where = tuple(slice(s.location[a], s.end_location[a]) for a in reversed(range(dim)))
        for name in data.files:
            part = data[name]
            lead = part.shape[:part.ndim - dim]
            if name not in out:
                out[name] = np.full(lead + box, np.nan, dtype=part.dtype)
            out[name][(slice(None),) * len(lead) + where] = part
    if save:
        np.savez(io.merged_filename(base, digits, it), **out)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--all', action='store_true', help='process every iteration of the series')
    ap.add_argument('sample', nargs='?')
    args = ap.parse_args(argv)
    if not args.sample:
        return 0
    base, sub_id, it, _ = args.sample.rsplit('.', 3)
    digits = len(it)
    if args.all:
        for fn in sorted(glob.glob('.'.join([base, sub_id, '[0-9]' * digits, 'npz']))):
            it = fn.rsplit('.', 3)[2]
            print('Processing {0}'.format(it))
            merge_subdomains(base, digits, int(it))
    else:
        merge_subdomains(base, digits, int(it))
    return 0


if __name__ == '__main__':
    sys.exit(main())
