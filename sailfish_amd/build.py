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
SOURCES = ['slf_kernels.hip', 'slf_slots.hip', 'slf_fast.hip', 'slf_row.hip', 'slf_sc.hip', 'slf_resident.hip', 'slf_api.hip']
HEADERS = ['slf_kernels.h', 'slf_lattice.h', 'slf_node.h', 'slf_sweep.h', 'slf_rowpush.h', os.path.join('..', '..', 'include', 'sailfish_hip.h')]

# -ffp-contract=off: fixed IEEE operation order (DESIGN.md "arithmetic contract");
# the sweep is HBM-bound, the extra VALU issue slots are hidden.
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
               '-Wno-unused-value']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def source_hash():
    """sha256 over every file under csrc/ (names and contents, sorted): what the kernels of a build were compiled from.
    Measurements that belong to one state of the kernels (profiles/traffic.json: PMC bytes per launch) carry it, and
    bench.py reports them only while it matches."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        path = os.path.join(CSRC, name)
        if os.path.isfile(path) and name.endswith(('.hip', '.h')):
            h.update(name.encode() + b'\0')
            with open(path, 'rb') as fh:
                h.update(fh.read())
            h.update(b'\0')
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=True):
    """Compile every HIP source into sailfish_amd/lib/libsailfish_hip.so (objects in parallel)."""
    if not force and not needs_build():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    procs = []
    objs = []
    cflags = [f for f in HIPCC_FLAGS if f != '-shared']
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(obj)
        st = os.path.getmtime(os.path.join(CSRC, src))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(st, hdr_t):
            continue
        cmd = [_hipcc()] + cflags + ['-c', '-o', obj, src]
        if verbose:
            print('[sailfish_amd.build]', ' '.join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIBPATH] + objs
    if verbose:
        print('[sailfish_amd.build]', ' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIBPATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
