"""Helpers of the weak-scaling harness (reference benchmark/scaling/util.py): result files with the
per-subdomain TimingInfo tuples and the total MLUPS (effective, compute-only)."""
import argparse
import os
import sys


def save_result(filename_base, num_blocks, timing_infos, min_timings, max_timings, subdomains):
    if int(os.environ.get('RANK', '0')) != 0:
        return None
    for tag, data in (('', timing_infos), ('_min', min_timings), ('_max', max_timings)):
        with open('%s%s_%d' % (filename_base, tag, num_blocks), 'w') as f:
            f.write(str(data))
    by_id = dict((s.id, s) for s in subdomains)
    eff = sum(by_id[ti.subdomain_id].num_nodes / ti.total * 1e-6 for ti in timing_infos)
    comp = sum(by_id[ti.subdomain_id].num_nodes / ti.comp * 1e-6 for ti in timing_infos if ti.comp > 0)
    with open('%s_mlups_%d' % (filename_base, num_blocks), 'w') as f:
        f.write('%.2f %.2f\n' % (eff, comp))
    return eff, comp


def process_cmdline():
    """--num_blocks N (default: WORLD_SIZE under torch.distributed.run, else 1); the remaining arguments
    go to the simulation's own option parser."""
    parser = argparse.ArgumentParser()
    parser.add_argument('--num_blocks', type=int, default=int(os.environ.get('WORLD_SIZE', '1')))
    parser.add_argument('--edge', type=int, default=512, help='subdomain edge (nodes) per block')
    args, remaining = parser.parse_known_args()
    del sys.argv[1:]
    sys.argv.extend(remaining)
    return args
