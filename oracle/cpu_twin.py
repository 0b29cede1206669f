"""ctypes front-end of oracle/lbm_fast.c: the blocked OpenMP twin of the periodic-box sweep, the CPU baseline that
bench.py reports.  TEST INFRASTRUCTURE ONLY -- never imported by sailfish_amd/.

The library is compiled on the machine it runs on (-O3 -march=native: the host CPU of the GPU box differs from the
build container's), once per CPU model, next to this file.
"""
import ctypes
import hashlib
import os
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}
_QUOTA = 'unset'


def cpu_model():
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def cpu_quota():
    """CPUs this process may actually use: the cgroup CPU bandwidth limit (v2 cpu.max or v1 cfs quota) and the
    affinity mask, whichever is smaller; None when unlimited / unknown.  A container that sees 256 logical CPUs but
    is allowed 8 CPUs' worth of time makes 128 pinned threads slower than 8."""
    limits = []
    try:
        limits.append(len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            quota, period = fh.read().split()[:2]
        if quota != 'max':
            limits.append(max(1, int(round(float(quota) / float(period)))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                limits.append(max(1, int(round(q / float(per)))))
        except (OSError, ValueError):
            pass
    return min(limits) if limits else None


def lib(precision=4):
    if precision in _libs:
        return _libs[precision]
    global _QUOTA
    if _QUOTA == 'unset':
        _QUOTA = cpu_quota()        # before libgomp pins this thread to one core
    # thread placement must be in the environment before libgomp initialises
    os.environ.setdefault('OMP_PLACES', 'cores')
    os.environ.setdefault('OMP_PROC_BIND', 'close')
    tag = hashlib.sha1(cpu_model().encode()).hexdigest()[:8]
    name = os.path.join(HERE, 'libfast_f%d_%s.so' % (precision * 8, tag))
    src = os.path.join(HERE, 'lbm_fast.c')
    if not os.path.exists(name) or os.path.getmtime(name) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O3', '-march=native', '-std=c99', '-fPIC', '-shared', '-ffp-contract=off',
                               '-fno-fast-math', '-fopenmp', '-DORC_REAL=%s' % ('float' if precision == 4 else 'double'),
                               '-o', name, src])
    L = ctypes.CDLL(name)
    assert L.fast_real_size() == precision
    vp = ctypes.c_void_p
    L.fast_run.argtypes = [ctypes.c_int] * 5 + [ctypes.c_double, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
    L.fast_run.restype = None
    L.fast_copy.argtypes = [ctypes.c_int] * 4 + [vp, vp]
    L.fast_copy.restype = None
    L.fast_set_threads.argtypes = [ctypes.c_int]
    _libs[precision] = L
    return L


class FastBox(object):
    """Fluid-only periodic box, AA pattern, BGK; arrays laid out like the oracle's ([Q, nz+2, ny+2, arr_nx])."""

    def __init__(self, lattice, size, visc, precision='single', alignment=32):
        self.lattice = 1 if lattice in (1, 'D3Q19') else 0
        self.dim = 3 if self.lattice else 2
        self.Q = 19 if self.lattice else 9
        self.size = tuple(size)
        self.nx, self.ny = size[0], size[1]
        self.nz = size[2] if self.dim == 3 else 1
        self.arr_nx = (self.nx + 2 + alignment - 1) // alignment * alignment
        self.prec = 4 if precision == 'single' else 8
        self.dtype = np.float32 if self.prec == 4 else np.float64
        self.tau = (6.0 * visc + 1.0) / 2.0          # sym.py:847-848
        self.L = lib(self.prec)
        self.shape = ((self.nz + 2) if self.dim == 3 else 1, self.ny + 2, self.arr_nx)
        self.dist = None
        self.iteration = 0

    def set_dist(self, dist):
        """Takes over an initial state (e.g. from the oracle's init): copied in parallel so that every thread first
        touches the rows it will sweep."""
        src = np.ascontiguousarray(dist, dtype=self.dtype).reshape((self.Q,) + self.shape)
        self.dist = np.empty_like(src)
        self.L.fast_copy(self.lattice, self.ny, self.nz, self.arr_nx, self.dist.ctypes.data, src.ctypes.data)
        self.iteration = 0

    def init_uniform(self, rho=1.0):
        """f_i = w_i rho everywhere (rest state), first-touched in parallel."""
        w = np.array([1 / 3.] + [1 / 18.] * 6 + [1 / 36.] * 12 if self.lattice else [4 / 9.] + [1 / 9.] * 4 + [1 / 36.] * 4)
        src = np.empty((self.Q,) + self.shape, dtype=self.dtype)
        for q in range(self.Q):
            src[q] = w[q] * rho
        self.set_dist(src)

    def run(self, steps, fields=None):
        f = [a.ctypes.data for a in fields] if fields is not None else [None] * 4
        if fields is not None and self.dim == 2:
            f = f[:3] + [None]
        self.L.fast_run(self.lattice, self.nx, self.ny, self.nz, self.arr_nx, self.tau, self.dist.ctypes.data,
                        self.iteration, steps, *f)
        self.iteration += steps

    def mlups(self, steps, repeats=3):
        """Best of `repeats` timings of `steps` (even) steps."""
        best = 0.0
        nodes = self.nx * self.ny * self.nz
        for _ in range(repeats):
            t0 = time.perf_counter()
            self.run(steps)
            dt = time.perf_counter() - t0
            best = max(best, nodes * steps / dt * 1e-6)
        return best


def baseline(model='bgk', precision='single', visc=1.0 / 6.0, budget_s=10.0):
    """The cpu_baseline object of bench.py: 256^3 D3Q19 (all cores and one thread) and BASELINE config 1
    (256^2 D2Q9), best of 3 each, threads pinned to cores."""
    L = lib(4 if precision == 'single' else 8)
    cores = L.fast_max_threads()
    quota = _QUOTA
    if quota is not None and quota < cores:
        cores = quota
        L.fast_set_threads(cores)
    note = ''
    if model != 'bgk':
        note = ' (the CPU twin implements BGK only; timed as BGK)'
    box = FastBox('D3Q19', (256, 256, 256), visc, precision)
    box.init_uniform()
    box.run(2)
    t0 = time.perf_counter()
    box.run(2)
    per2 = time.perf_counter() - t0
    steps = max(2, int(budget_s * 0.45 / 3 / per2) * 2)
    all_cores = box.mlups(steps)
    L.fast_set_threads(1)
    one = box.mlups(2, repeats=1)
    L.fast_set_threads(cores)
    # the pure-NumPy twin "for context" (SURVEY.md section 8(d)): the kind of CPU path a Python reference would have had
    from oracle import numpy_twin
    from sailfish_amd import sym
    np_mlups = numpy_twin.mlups(sym.D3Q19, (64, 64, 64), visc, steps=2)
    c1 = FastBox('D2Q9', (256, 256), 0.0254, precision)
    c1.init_uniform()
    c1.run(20)
    c1_mlups = c1.mlups(400)
    return {'value': round(all_cores, 1), 'unit': 'MLUPS', 'cores': cores, 'kind': 'port',
            'logical_cpus': os.cpu_count(), 'cpu_quota': quota,
            'single_thread_mlups': round(one, 2), 'config1_d2q9_256x256_mlups': round(c1_mlups, 1),
            'numpy_twin_mlups': round(np_mlups, 2),
            'mlups_per_thread': round(all_cores / cores, 2),
            'sample': 'oracle/lbm_fast.c (OpenMP, %d threads pinned to cores, %s): D3Q19 BGK f%d AA periodic 256^3, best of 3 x '
                      '%d steps; one thread: 2 steps; config 1 = D2Q9 256^2, best of 3 x 400 steps; numpy_twin = oracle/numpy_twin.py '
                      '(np.roll, f64, one thread) D3Q19 64^3 x 2 steps%s'
                      % (cores, cpu_model(), 32 if precision == 'single' else 64, steps, note)}
