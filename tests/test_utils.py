"""utils/merge_subdomains.py and utils/compare_results.py (tools of the reference) on
synthetic per-subdomain files."""
import pickle
import subprocess
import sys
import os

import numpy as np

from sailfish_amd import io
from sailfish_amd.subdomain import SubdomainSpec2D, SubdomainSpec3D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_merge_3d(tmp_path):
    base = str(tmp_path / 'run')
    specs = [SubdomainSpec3D((0, 0, 0), (6, 5, 4), envelope_size=1, id_=0),
             SubdomainSpec3D((6, 0, 0), (3, 5, 4), envelope_size=1, id_=1)]
    with open(io.subdomains_filename(base), 'wb') as f:
        pickle.dump(specs, f)
    rng = np.random.RandomState(3)
    full_rho = rng.rand(4, 5, 9).astype(np.float32)
    full_v = rng.rand(3, 4, 5, 9).astype(np.float32)
    for s in specs:
        x0, x1 = s.location[0], s.end_location[0]
        np.savez(io.filename(base, 3, s.id, 20), rho=full_rho[:, :, x0:x1], v=full_v[:, :, :, x0:x1])
    rc = subprocess.call([sys.executable, os.path.join(ROOT, 'utils', 'merge_subdomains.py'),
                          io.filename(base, 3, 0, 20)])
    assert rc == 0
    merged = np.load(io.merged_filename(base, 3, 20))
    assert np.array_equal(merged['rho'], full_rho) and np.array_equal(merged['v'], full_v)


def test_merge_2d_with_hole_and_compare(tmp_path):
    base = str(tmp_path / 'r2')
    specs = [SubdomainSpec2D((0, 0), (4, 4), envelope_size=1, id_=0),
             SubdomainSpec2D((4, 2), (4, 2), envelope_size=1, id_=1)]
    with open(io.subdomains_filename(base), 'wb') as f:
        pickle.dump(specs, f)
    np.savez(io.filename(base, 2, 0, 5), rho=np.ones((4, 4), np.float32))
    np.savez(io.filename(base, 2, 1, 5), rho=2 * np.ones((2, 4), np.float32))
    sys.path.insert(0, ROOT)
    from utils.merge_subdomains import merge_subdomains
    from utils.compare_results import compare
    out = merge_subdomains(base, 2, 5)
    assert out['rho'].shape == (4, 8)
    assert np.all(out['rho'][:, :4] == 1) and np.all(out['rho'][2:, 4:] == 2) and np.all(np.isnan(out['rho'][:2, 4:]))
    m = io.merged_filename(base, 2, 5)
    assert compare(m, m, out=open(os.devnull, 'w')) == 0        # NaN == NaN
    other = str(tmp_path / 'other.npz')
    rho2 = out['rho'].copy()
    rho2[3, 3] += 1e-3
    np.savez(other, rho=rho2)
    assert compare(m, other, out=open(os.devnull, 'w')) == 1
    rc = subprocess.call([sys.executable, os.path.join(ROOT, 'utils', 'compare_results.py'), m, other],
                         stderr=subprocess.DEVNULL)
    assert rc == 1
