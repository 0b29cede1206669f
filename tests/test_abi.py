"""The drop-in boundary on a machine without a GPU: libsailfish_hip.so loads, exports every entry
point include/sailfish_hip.h declares, the ctypes mirror of slf_module_desc has the C layout, the
product fails loudly without the library, and the process ends up with ONE HIP runtime whatever
the import order of torch and the backend.  No compute call is made."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from sailfish_amd import hipabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'sailfish_hip.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'^\s*(?:const\s+)?(?:int|char\s*\*|void)\s*\*?\s*(slf_[a-z0-9_]+)\s*\(', text, flags=re.M)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 36 and 'slf_kernel_launch' in names and 'slf_last_error' in names
    lib = hipabi.load()
    for n in names:
        assert hasattr(lib, n), 'libsailfish_hip.so does not export %s' % n
        assert n in hipabi.SIGNATURES, 'hipabi.SIGNATURES does not bind %s' % n
    assert sorted(hipabi.SIGNATURES) == names
    assert lib.slf_abi_version() == 1


def test_module_desc_layout_matches_the_header(tmp_path):
    fields = [f[0] for f in hipabi.SlfModuleDesc._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sailfish_hip.h"\nint main(void) {\n'
                   '  printf("%zu\\n", sizeof(slf_module_desc));\n' +
                   ''.join('  printf("%%zu\\n", offsetof(slf_module_desc, %s));\n' % f for f in fields) +
                   '  printf("%zu\\n", sizeof(slf_region));\n  return 0;\n}\n')
    exe = str(tmp_path / 'layout')
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', exe, str(src)])
    out = [int(x) for x in subprocess.check_output([exe]).split()]
    assert out[0] == ctypes.sizeof(hipabi.SlfModuleDesc)
    for f, off in zip(fields, out[1:-1]):
        assert getattr(hipabi.SlfModuleDesc, f).offset == off, f
    assert out[-1] == ctypes.sizeof(hipabi.SlfRegion)


def test_no_cpu_fallback():
    with pytest.raises(hipabi.HipLibraryMissing):
        hipabi.load('/nonexistent/libsailfish_hip.so')


def test_errors_are_reported_not_swallowed():
    lib = hipabi.load()
    rc = lib.slf_ctx_create(0, None)
    assert rc != 0 and b'NULL' in lib.slf_last_error()


@pytest.mark.parametrize('order', ['backend_first', 'torch_first'])
def test_single_hip_runtime(order):
    """torch bundles its own libamdhip64; the backend must bind to that copy (hipabi._share_hip_runtime_with_torch)
    or streams could not be shared with torch.distributed / RCCL."""
    code = {'backend_first': 'from sailfish_amd import hipabi; hipabi.load(); import torch',
            'torch_first': 'import torch; from sailfish_amd import hipabi; hipabi.load()'}[order]
    code += ("\nimport re\nm = open('/proc/self/maps').read()\n"
             "print(len(set(re.findall(r'/\\S*libamdhip64[^\\s]*', m))), len(set(re.findall(r'/\\S*libhsa-runtime64[^\\s]*', m))))")
    out = subprocess.check_output([sys.executable, '-c', code], cwd=ROOT).split()
    assert [int(x) for x in out[-2:]] == [1, 1]


# every in-scope `${kernel}` of sailfish/templates/, under the reference's own name
REFERENCE_KERNELS = [
    'CollideAndPropagate', 'SetInitialConditions', 'PrepareMacroFields', 'ApplyPeriodicBoundaryConditions',
    'ApplyPeriodicBoundaryConditionsWithSwap', 'ApplyMacroPeriodicBoundaryConditions', 'CollectContinuousData',
    'CollectContinuousDataWithSwap', 'DistributeContinuousData', 'DistributeContinuousDataWithSwap', 'CollectSparseData',
    'DistributeSparseData', 'CollectContinuousMacroData', 'DistributeContinuousMacroData', 'ShanChenPrepareMacroFields',
    'ShanChenCollideAndPropagate0', 'ShanChenCollideAndPropagate1']


def test_every_in_scope_reference_kernel_name_is_served():
    """slf_kernel_get() resolves names with a chain of string compares: every name of the table must be in it (the GPU
    round trips of the face kernels are tests/test_gpu_face_kernels.py; the sweeps and PBC kernels run in every test)."""
    src = open(os.path.join(ROOT, 'sailfish_amd', 'csrc', 'slf_api.hip')).read()
    served = set(re.findall(r'!strcmp\(name, "([A-Za-z0-9]+)"\)', src))
    missing = [n for n in REFERENCE_KERNELS if n not in served]
    assert not missing, missing
    header = open(HEADER).read()
    for n in REFERENCE_KERNELS:
        assert n.replace('0', '').replace('1', '') in header.replace('0|1', ''), 'include/sailfish_hip.h does not document %s' % n


# kernels this library adds to the reference's set (documented in include/sailfish_hip.h next to the ones they stand in for)
ADDED_KERNELS = ['ShanChenCollideAndPropagateFused', 'ShanChenCollideAndPropagateFusedV', 'ShanChenPrepareDensities',
                 'ComputeMacroFields', 'CollideAndPropagateResident']


def test_added_kernel_names_are_served_and_documented():
    src = open(os.path.join(ROOT, 'sailfish_amd', 'csrc', 'slf_api.hip')).read()
    served = set(re.findall(r'!strcmp\(name, "([A-Za-z0-9]+)"\)', src))
    header = open(HEADER).read()
    for n in ADDED_KERNELS:
        assert n in served, n
        assert n in header, 'include/sailfish_hip.h does not document %s' % n
    # and nothing is served that neither list knows
    assert served == set(REFERENCE_KERNELS) | set(ADDED_KERNELS), served ^ (set(REFERENCE_KERNELS) | set(ADDED_KERNELS))
