"""Ahead-of-time build of libsailfish_hip.so for gfx950 (MI355X).

Replaces the reference's run-time code generation + compilation
(sailfish/codegen.py:104-180 -> backend_cuda.py:193-218): kernels are
pre-written HIP, compiled once with hipcc, in-tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIBPATH = os.path.join(LIBDIR, 'libsailfish_hip.so')
SOURCES = ['slf_kernels.hip', 'slf_api.hip']
HEADERS = ['slf_kernels.h', 'slf_lattice.h', 'slf_node.h', os.path.join('..', '..', 'include', 'sailfish_hip.h')]

# -ffp-contract=off: fixed IEEE operation order (DESIGN.md "arithmetic contract");
# the sweep is HBM-bound, the extra VALU issue slots are hidden.
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
               '-Wno-unused-value']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=True):
    """Compile every HIP source into sailfish_amd/lib/libsailfish_hip.so."""
    if not force and not needs_build():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [_hipcc()] + HIPCC_FLAGS + ['-o', LIBPATH] + SOURCES
    if verbose:
        print('[sailfish_amd.build]', ' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIBPATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
