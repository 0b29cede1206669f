"""x faces connected to another subdomain without ghost-column traffic (C ABI: slf_module_set_xface_buffers).

A workgroup owns whole rows, so an x face is one node per row: pushing into the ghost columns costs a partial-line
write per row and direction, packing / unpacking them a strided gather / scatter (64-byte sectors for 4-byte values).
Here the two edge lanes of every row write what leaves the subdomain into dense send buffers and read what enters it
from dense receive buffers ([k][z][y] over the padded plane, k = rank of the direction among the 5 with e_x > 0 resp.
< 0); per step the host moves send_high -> the high neighbour's recv_low and send_low -> the low neighbour's
recv_high, nothing else.  The distribution arrays are then stale at the face until `materialise()` copies the
receive buffers into them (before a checkpoint / a debug dump); receive buffers full of NaN mean "the arrays count".

Replaces, for 1-D decompositions along x (the reference's default axis, geo.py:100-135), the reference's
CollectContinuousData / DistributeContinuousData on x faces (kernel_utils.mako:526-543, 692-708).
"""
import ctypes

import numpy as np

from sailfish_amd import sym

LOW, HIGH = 0, 1


def supported(grid, desc, indirect=False, simtype=0):
    return grid.dim == 3 and grid.Q == 19 and not indirect and not simtype and desc.lat_nx - 2 <= 1024


def face_count(desc):
    """Elements of one face buffer: 5 directions x the padded (arr_ny x arr_nz) plane."""
    return 5 * desc.arr_ny * desc.arr_nz


class XFaceHalo(object):
    def __init__(self, backend, module, grid, desc, send, recv):
        """send / recv: device addresses [low face, high face] of buffers of face_count(desc) reals each, 0 for a
        face that is not connected."""
        self.backend, self.module, self.grid, self.desc = backend, module, grid, desc
        self.dtype = np.float32 if desc.precision == 4 else np.float64
        self.nrows = desc.arr_ny * desc.arr_nz
        self.count = face_count(desc)
        self.nbytes = self.count * self.dtype().itemsize
        self.send, self.recv = list(send), list(recv)
        # directions entering through the low face have e_x > 0, through the high face e_x < 0 (ascending = rank order)
        self.enter = [sym.get_prop_dists(grid, 1, 0), sym.get_prop_dists(grid, -1, 0)]
        self._kernels = {}
        lib = backend._lib
        from sailfish_amd.backend_hip import _check
        _check(lib, lib.slf_module_set_xface_buffers(module.handle, *[ctypes.c_void_p(a or None) for a in
                                                                      (self.send[LOW], self.send[HIGH],
                                                                       self.recv[LOW], self.recv[HIGH])]),
               'slf_module_set_xface_buffers')

    @classmethod
    def allocate(cls, backend, module, grid, desc, faces, alloc):
        """faces: (low connected, high connected); alloc(n_elements) -> device address of a buffer the transport can
        send from / receive into.  Allocation order: send low, send high, receive low, receive high."""
        n = face_count(desc)
        send = [alloc(n) if faces[f] else 0 for f in (LOW, HIGH)]
        recv = [alloc(n) if faces[f] else 0 for f in (LOW, HIGH)]
        return cls(backend, module, grid, desc, send, recv)

    def reset(self, stream=None):
        """All entries NaN: nothing has crossed the faces yet, the kernels read the arrays."""
        for a in self.send + self.recv:
            if a:
                self.backend.memset_buf(a, 0xFF, self.nbytes, stream)

    @property
    def needs_clear(self):
        """Without in-sweep wrap along y and z some rows of the send buffers are not written in every step (nothing
        is pushed from a ghost row); they must read as 'nothing crossed here' (NaN), not as last step's value."""
        d = self.desc
        return not (d.periodic_fused[1] and d.periodic_fused[2])

    def clear_send(self, stream):
        for a in self.send:
            if a:
                self.backend.memset_buf(a, 0xFF, self.nbytes, stream)

    def materialise(self, dist, pushed, stream):
        """Writes the receive buffers into the distribution array `dist`: pushed = True after a push step (AB, odd AA:
        the values belong into the first real column, same slots), False after the even AA step (they belong into the
        ghost column, opposite slots, where the next pull looks for them)."""
        b, d = self.backend, self.desc
        nx = d.lat_nx - 2
        isz = self.dtype().itemsize
        for face in (LOW, HIGH):
            if not self.recv[face]:
                continue
            if pushed:
                x = 1 if face == LOW else nx
                mask = 0
                for q in self.enter[face]:
                    mask |= 1 << q
                jobs = [(mask, x, 0)]
            else:
                x = 0 if face == LOW else nx + 1
                jobs = [(1 << self.grid.idx_opposite[q], x, k * self.nrows * isz) for k, q in enumerate(self.enter[face])]
            for mask, col, off in jobs:
                key = (dist, face, pushed, mask)
                if key not in self._kernels:
                    self._kernels[key] = b.get_kernel(self.module, 'DistributeContinuousData', (64,),
                                                      [dist, self.recv[face] + off, mask, col, d.arr_nx, d.arr_ny,
                                                       d.arr_nx * d.arr_ny, d.arr_nz], 'PPiiiiii')
                b.run_kernel(self._kernels[key], None, stream)
