"""Worker of tests/test_gpu_comm.py: one-rank RCCL communicator through the C ABI (slf_comm_*), in its own process."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from sailfish_amd.backend_hip import HIPBackend, _check
mode = sys.argv[1]
class Opt(object): pass
b = HIPBackend(Opt(), 0)
lib = b._lib
uid = ctypes.create_string_buffer(128)
_check(lib, lib.slf_comm_unique_id(uid), 'uid')
comm = ctypes.c_void_p()
_check(lib, lib.slf_comm_init(b._ctx, 1, 0, uid, ctypes.byref(comm)), 'init')
cn, cr = ctypes.c_int(-1), ctypes.c_int(-1)
_check(lib, lib.slf_comm_count(comm, ctypes.byref(cn), ctypes.byref(cr)), 'count')     # ncclCommCount / ncclCommUserRank
print(mode, 'count ok', (cn.value, cr.value) == (1, 0), flush=True)
n = 5 * 512 * 512                      # one x-face of the 8-GPU layout
src = np.arange(n, dtype=np.float32)
g_src = b.alloc_buf(like=src); g_dst = b.alloc_buf(size=n * 4)
stream = b.make_stream()
for _ in range(3):
    _check(lib, lib.slf_comm_group_begin(), 'group_begin')
    _check(lib, lib.slf_comm_sendrecv(comm, 0, ctypes.c_void_p(g_src), n, ctypes.c_void_p(g_dst), n, 4, stream.handle), 'sr')
    _check(lib, lib.slf_comm_group_end(), 'group_end')
stream.synchronize()
out = np.zeros(n, dtype=np.float32); b.from_buf(g_dst, out)
print(mode, 'data ok', np.array_equal(out, src), flush=True)
if mode == 'destroy':
    _check(lib, lib.slf_comm_destroy(comm), 'destroy')
elif mode == 'sync_destroy':
    b.sync(); _check(lib, lib.slf_comm_destroy(comm), 'destroy')
elif mode == 'leak':
    pass
elif mode == 'destroy_exit0':
    _check(lib, lib.slf_comm_destroy(comm), 'destroy'); sys.stdout.flush(); os._exit(0)
print(mode, 'end of script', flush=True)
