// gfx950 (MI355X, CDNA4) kernels of the Sailfish hot path.
//
//   sweep_kernel      fused collide-and-stream  = reference CollideAndPropagate
//                     (sailfish/templates/models/lb_single_fluid.mako:161-229)
//   init_kernel       SetInitialConditions      (lb_single_fluid.mako:101-127)
//   pbc_kernel        ApplyPeriodicBoundaryConditions[WithSwap] (kernel_utils.mako:19-384)
//   macro_pbc_kernel  ApplyMacroPeriodicBoundaryConditions      (kernel_utils.mako:410-474)
//   sparse_kernel     Collect/DistributeSparseData              (kernel_utils.mako:796-836)
//
// Design: bandwidth-bound stencil, no MFMA.  SoA layout
// dist[q][z][y][x] with x padded (reference subdomain_runner.py:367-373), one
// thread per node along x so that every population access of a wave is one
// contiguous 256-byte row segment; a workgroup owns a contiguous x-chunk of one
// (y, z) row, so y/z neighbour offsets (and the periodic wrap of those axes) are
// wave-uniform scalars and only the +-1 x shift is per lane.  Streaming is
// "push" for AB and the odd AA step, in-place opposite-slot for the even AA step
// (reference propagation.mako:170-174, 384-421; geo_helpers.mako:248-276).
#include "../../include/sailfish_hip.h"
#include "slf_kernels.h"
#include "slf_node.h"
#include "slf_sweep.h"

namespace slf {

// TURB (--regularized / --subgrid): workgroups of at most 512 threads, i.e. 256 VGPRs -- the non-equilibrium flux tensor on
// top of the collision does not fit the 128 of a 1024-thread workgroup (76-152 bytes of scratch per lane in the odd step).
template <class L, class R, int MODEL, int PROP, bool GENERAL, bool INDIRECT = false, bool ROUNDOFF = false, bool TURB = false>
__global__ void __launch_bounds__(TURB ? 512 : 1024) sweep_kernel(const SweepParams<L, R> p) {
  const Geometry& g = p.g;
  const int gy = p.y0 + (int)blockIdx.y;
  const int gz = (L::dim == 3) ? p.z0 + (int)blockIdx.z : 0;
  // thread 0 of a row owns the first *real* node (x = 1): with the distribution arrays allocated so that
  // x = 1 sits on a 128-byte boundary (backend alloc_buf(align_offset=...)), every wave's row segment
  // is line aligned
  const int gx = 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (gx > g.lat_nx - 2) return;

  const uint32_t gi = (uint32_t)gx + (uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz;

  // indirect addressing (reference kernel_common.mako:140-167): the distribution arrays hold active
  // nodes only; si = slot of this node, neighbours are translated through the same dense table
  uint32_t si = gi;
  if constexpr (INDIRECT) {
    si = p.nodes[gi];
    if (si == INVALID_NODE) return;
  }
  int kind = NK_FLUID;
  uint32_t code = 0;
  if constexpr (GENERAL) {
    code = p.map[gi];
    kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
    if (kind_is_excluded(kind)) return;
  }

  const AxisOff ox = axis_off(gx, g.lat_nx, 1, g.wrap[0]);
  const AxisOff oy = axis_off(gy, g.lat_ny, g.arr_nx, g.wrap[1]);
  const AxisOff oz = (L::dim == 3) ? axis_off(gz, g.lat_nz, g.arr_nxy, g.wrap[2]) : AxisOff{0, 0};
  const size_t ds = g.dist_size;

  // ---- load (reference getDist, geo_helpers.mako:258-276)
  R f[L::Q];
  static_for<0, L::Q>([&](auto I) {
    if constexpr (PROP == PROP_AA_ODD) {
      const int off = dir_offset<L, I>(ox, oy, oz, false);
      uint32_t sn = (uint32_t)((int)gi + off);
      if constexpr (INDIRECT) sn = p.nodes[sn];
      f[I] = (!INDIRECT || sn != INVALID_NODE) ? (p.din + ds * (size_t)L::opp(I))[sn] : (R)0;
    } else {
      f[I] = (p.din + ds * (size_t)I)[si];
    }
  });

  R rho, v[3];
  bool wet = true;
  node_update<L, R, MODEL, PROP, GENERAL, INDIRECT, FORCE_RUNTIME, 2, ROUNDOFF, TURB>(p, f, code, kind, gi, ox, oy, oz, rho, v, wet, si);

  if (wet) check_invalid<R>(p.status, p.options, rho, gx, gy, gz);
  // ---- macroscopic output (save_macro_fields, kernel_common.mako:213-240)
  if ((p.options & 1u) && wet) {
    p.rho[gi] = rho;
    p.vx[gi] = v[0];
    p.vy[gi] = v[1];
    if constexpr (L::dim == 3) p.vz[gi] = v[2];
  }

  // ---- streaming (propagate, propagation.mako:384-421)
  static_for<0, L::Q>([&](auto I) {
    if constexpr (PROP == PROP_AA_EVEN) {
      (p.dout + ds * (size_t)L::opp(I))[si] = f[I];
    } else {
      const int off = dir_offset<L, I>(ox, oy, oz, true);
      uint32_t t = (uint32_t)((int)gi + off);
      if constexpr (INDIRECT) t = p.nodes[t];
      if (!INDIRECT || t != INVALID_NODE) (p.dout + ds * (size_t)I)[t] = f[I];
    }
  });
}

// SetInitialConditions: f_i = feq_i(rho, v) on *every* node of the lattice box,
// no type test (ghost nodes carry the non-finite sentinels of the host fields).
template <class L, class R>
__global__ void __launch_bounds__(256) init_kernel(R* dist, const R* __restrict__ irho, const R* __restrict__ ivx,
                                                   const R* __restrict__ ivy, const R* __restrict__ ivz, Geometry g,
                                                   int incompressible, const uint32_t* __restrict__ nodes) {
  const int gx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int gy = (int)blockIdx.y;
  const int gz = (int)blockIdx.z;
  if (gx > g.lat_nx - 1) return;
  const uint32_t gi = (uint32_t)gx + (uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz;
  uint32_t si = gi;   // indirect addressing: slot of this node, if it is active
  if (nodes) {
    si = nodes[gi];
    if (si == INVALID_NODE) return;
  }
  // incompressible = the module's density model: SLF_DENSITY_ROUNDOFF stores f_i - w_i (lb_single_fluid.mako:113)
  const R rho = (incompressible == SLF_DENSITY_ROUNDOFF) ? irho[gi] - (R)1 : irho[gi];
  R v[3];
  v[0] = ivx[gi];
  v[1] = ivy[gi];
  v[2] = (R)0;
  if constexpr (L::dim == 3) v[2] = ivz[gi];
  const R rho0 = (incompressible == SLF_DENSITY_ROUNDOFF) ? rho + (R)1 : (incompressible ? (R)1 : rho);
  const R u15 = usq15<L, R>(v);
  static_for<0, L::Q>([&](auto I) { (dist + (size_t)g.dist_size * (size_t)I)[si] = feq<L, R, I>(rho, rho0, v, u15); });
}

__device__ __forceinline__ bool slf_isfinite(float x) { return __builtin_isfinite(x); }
__device__ __forceinline__ bool slf_isfinite(double x) { return __builtin_isfinite(x); }

// Which face nodes take part in the PBC copy of `axis`, judged along another axis b
// (coordinate c, population component e, mode of b: 0 not locally periodic, 1 ghost-layer
// PBC, 2 wrapped in-sweep; `earlier` = b is handled before `axis` in the x, y, z order).
//
// After a push step a ghost slot holds data only if it was pushed from a real node:
// e = +1 -> c >= 2, e = -1 -> c <= lat-3.  Along a PBC axis handled earlier the wrapped
// copies already sit on real nodes; along a later PBC axis the ghost rows still hold the
// edge populations (they are moved on by that axis' kernel).  Ghost rows of axes that are
// not locally periodic belong to walls or to other subdomains (halo exchange) and never
// take part; along an in-sweep wrapped axis there are no ghost rows at all.
__device__ __forceinline__ bool pbc_push_ok(int c, int e, int lat, int mode, bool earlier) {
  const bool real = c >= 1 && c <= lat - 2;
  if (mode == 2) return real;
  if (mode == 1 && earlier) return real;
  const bool pushed = (e > 0) ? c >= 2 : ((e < 0) ? c <= lat - 3 : real);
  if (mode == 1) return pushed;
  return real && pushed;
}

// Swap variant (ghost <- opposite real layer, for the pull of the next step): along a PBC
// axis handled earlier the already filled ghost columns take part when the pulling node
// (at c + e) is real; everything else contributes its real range only.
__device__ __forceinline__ bool pbc_pull_ok(int c, int e, int lat, int mode, bool earlier) {
  if (mode == 1 && earlier) return c + e >= 1 && c + e <= lat - 2;
  return c >= 1 && c <= lat - 2;
}

// Periodic boundary conditions inside one subdomain for one axis.
//
// !SWAP (after a push step; effect of kernel_utils.mako:295-336): populations
// that were pushed into the ghost layer of `axis` are moved to the real layer on
// the opposite side.  Applied in x, y, z order this moves edge/corner populations
// one axis at a time (see pbc_push_ok for which face nodes take part);
// non-finite values (never-written ghost slots) are skipped.
//
// SWAP (after the in-place even AA step; effect of kernel_utils.mako:343-384):
// real layer -> opposite ghost layer, opposite slots, so that the next (odd)
// step can pull through the periodic face.  Axes already processed (lower
// index) contribute their ghost columns, later axes only their real range.
template <class L, class R, bool SWAP>
__global__ void __launch_bounds__(256) pbc_kernel(R* dist, Geometry g, int axis) {
  const int lat[3] = {g.lat_nx, g.lat_ny, g.lat_nz};
  const int stride[3] = {1, g.arr_nx, g.arr_nxy};
  int b_ax, c_ax;  // the two other axes, b fastest
  if (axis == 0) { b_ax = 1; c_ax = 2; }
  else if (axis == 1) { b_ax = 0; c_ax = 2; }
  else { b_ax = 0; c_ax = 1; }
  const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int c = (L::dim == 3) ? (int)blockIdx.y : 0;
  if (b > lat[b_ax] - 1) return;
  if (L::dim == 2) c_ax = 2;  // lat_nz == 1, c == 0

  const int n = lat[axis];
  const size_t ds = g.dist_size;
  const uint32_t base = (uint32_t)b * (uint32_t)stride[b_ax] + (uint32_t)c * (uint32_t)stride[c_ax];

  static_for<1, L::Q>([&](auto I) {
    const int e[3] = {L::ex(I), L::ey(I), L::ez(I)};
    const int ea = e[axis];
    if (ea == 0) return;
    const int eb = e[b_ax];
    const int ec = (L::dim == 3) ? e[c_ax] : 0;
    if constexpr (!SWAP) {
      if (!pbc_push_ok(b, eb, lat[b_ax], g.axis_mode[b_ax], b_ax < axis)) return;
      if (L::dim == 3) {
        if (!pbc_push_ok(c, ec, lat[c_ax], g.axis_mode[c_ax], c_ax < axis)) return;
      }
      // ea = -1: landed in the low ghost (0) -> high real (n-2); ea = +1: high ghost (n-1) -> low real (1)
      const uint32_t src = base + (uint32_t)((ea < 0 ? 0 : n - 1) * stride[axis]);
      const uint32_t dst = base + (uint32_t)((ea < 0 ? n - 2 : 1) * stride[axis]);
      R* d = dist + ds * (size_t)I;
      const R val = d[src];
      if (slf_isfinite(val)) d[dst] = val;
    } else {
      if (!pbc_pull_ok(b, eb, lat[b_ax], g.axis_mode[b_ax], b_ax < axis)) return;
      if (L::dim == 3) {
        if (!pbc_pull_ok(c, ec, lat[c_ax], g.axis_mode[c_ax], c_ax < axis)) return;
      }
      // ea = +1: puller at 1 reads ghost 0 <- real n-2 ; ea = -1: puller at n-2 reads ghost n-1 <- real 1
      const uint32_t src = base + (uint32_t)((ea > 0 ? n - 2 : 1) * stride[axis]);
      const uint32_t dst = base + (uint32_t)((ea > 0 ? 0 : n - 1) * stride[axis]);
      R* d = dist + ds * (size_t)L::opp(I);
      const R val = d[src];
      if (slf_isfinite(val)) d[dst] = val;
    }
  });
}

// Ghost fill of a scalar field for one periodic axis (kernel_utils.mako:410-474).
template <class R>
__global__ void __launch_bounds__(256) macro_pbc_kernel(R* field, Geometry g, int axis) {
  const int lat[3] = {g.lat_nx, g.lat_ny, g.lat_nz};
  const int stride[3] = {1, g.arr_nx, g.arr_nxy};
  int b_ax, c_ax;
  if (axis == 0) { b_ax = 1; c_ax = 2; }
  else if (axis == 1) { b_ax = 0; c_ax = 2; }
  else { b_ax = 0; c_ax = 1; }
  const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int c = (int)blockIdx.y;
  if (b > lat[b_ax] - 1 || c > lat[c_ax] - 1) return;
  const int n = lat[axis];
  const uint32_t base = (uint32_t)b * (uint32_t)stride[b_ax] + (uint32_t)c * (uint32_t)stride[c_ax];
  const R hi = field[base + (uint32_t)((n - 2) * stride[axis])];
  if (slf_isfinite(hi)) field[base] = hi;
  const R lo = field[base + (uint32_t)stride[axis]];
  if (slf_isfinite(lo)) field[base + (uint32_t)((n - 1) * stride[axis])] = lo;
}

// Index-list gather / scatter of populations (halo pack / unpack).
template <class R, bool COLLECT>
__global__ void __launch_bounds__(256) sparse_kernel(const unsigned long long* __restrict__ idx, R* dist, R* buffer,
                                                     int n) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const unsigned long long gi = idx[i];
  if (gi == ~0ull) return;
  if constexpr (COLLECT) {
    buffer[i] = dist[gi];
  } else {
    // never-written ghost slots (pushed from excluded nodes) stay non-finite sentinels and are not
    // delivered -- the same policy as the periodic-boundary kernels (reference kernel_utils.mako:196,229)
    const R val = buffer[i];
    if (slf_isfinite(val)) dist[gi] = val;
  }
}

// Face pack / unpack without index lists (reference Collect/DistributeContinuousData, kernel_utils.mako:526-543,
// 629-645, 692-708, 777-793): the populations `dirs` (bit mask over q, ascending) of an ncols x nrows box of nodes
// base + c * col_stride + r * row_stride  <->  dense buffer [k][r][c].  z and y faces have col_stride = 1: every
// wave moves whole contiguous row segments; x faces (col_stride = arr_nx) are a strided gather / scatter, still
// without the 8 bytes of index per 4 bytes of payload of the sparse kernels.
template <class R, bool COLLECT>
__global__ void __launch_bounds__(256) box_kernel(R* dist, R* buffer, size_t dq, unsigned long long dirlist, unsigned long long base,
                                                  long long col_stride, int ncols, long long row_stride, int nrows,
                                                  long long buf_k_stride, long long buf_row_stride, int deliver_all) {
  const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int r = (int)blockIdx.y;
  if (c >= ncols) return;
  const int q = (int)((dirlist >> (5u * blockIdx.z)) & 31ull);  // the blockIdx.z-th direction of the list (5 bits each)
  R* node = dist + dq * (size_t)q + base + (size_t)((long long)c * col_stride + (long long)r * row_stride);
  R* slot = buffer + (size_t)blockIdx.z * (size_t)buf_k_stride + (size_t)r * (size_t)buf_row_stride + (size_t)c;
  if constexpr (COLLECT) {
    *slot = *node;
  } else {
    const R val = *slot;
    if (deliver_all || slf_isfinite(val)) *node = val;   // populations: sentinels are not delivered (as sparse_kernel / the PBC kernels)
  }
}

// dirs: the directions whose layers travel, in buffer order (at most 12, 5 bits each in the kernel argument).
hipError_t launch_box(const KernelSelector& sel, const Geometry& g, bool collect, void* dist, void* buffer,
                      const int* dirs, int nd, unsigned long long base, long long col_stride, int ncols, long long row_stride,
                      int nrows, long long buf_k_stride, long long buf_row_stride, bool deliver_all, hipStream_t s) {
  if (buf_k_stride <= 0) buf_k_stride = (long long)nrows * ncols;     // dense [k][r][c]
  if (buf_row_stride <= 0) buf_row_stride = ncols;
  if (nd <= 0 || ncols <= 0 || nrows <= 0) return hipSuccess;
  if (nd > 12) return hipErrorInvalidValue;
  unsigned long long dirlist = 0;
  for (int i = 0; i < nd; i++) dirlist |= (unsigned long long)(dirs[i] & 31) << (5 * i);
  dim3 block(256, 1, 1);
  dim3 grid((ncols + 255) / 256, nrows, nd);
  const size_t dq = g.dist_size;
  const int all = deliver_all ? 1 : 0;
  if (sel.precision == 4) {
    if (collect) hipLaunchKernelGGL((box_kernel<float, true>), grid, block, 0, s, (float*)dist, (float*)buffer, dq, dirlist, base, col_stride, ncols, row_stride, nrows, buf_k_stride, buf_row_stride, all);
    else hipLaunchKernelGGL((box_kernel<float, false>), grid, block, 0, s, (float*)dist, (float*)buffer, dq, dirlist, base, col_stride, ncols, row_stride, nrows, buf_k_stride, buf_row_stride, all);
  } else {
    if (collect) hipLaunchKernelGGL((box_kernel<double, true>), grid, block, 0, s, (double*)dist, (double*)buffer, dq, dirlist, base, col_stride, ncols, row_stride, nrows, buf_k_stride, buf_row_stride, all);
    else hipLaunchKernelGGL((box_kernel<double, false>), grid, block, 0, s, (double*)dist, (double*)buffer, dq, dirlist, base, col_stride, ncols, row_stride, nrows, buf_k_stride, buf_row_stride, all);
  }
  return hipGetLastError();
}

// PrepareMacroFields-style pass: density/velocity of every wet node without
// collision or streaming (lb_single_fluid.mako:129-159, used for output of
// the current state, e.g. right after initialisation or a restore).
template <class L, class R, int PROP, bool GENERAL>
__global__ void __launch_bounds__(1024) macro_kernel(const SweepParams<L, R> p) {
  const Geometry& g = p.g;
  const int gy = p.y0 + (int)blockIdx.y;
  const int gz = (L::dim == 3) ? p.z0 + (int)blockIdx.z : 0;
  const int gx = 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (gx > g.lat_nx - 2) return;
  const uint32_t gi = (uint32_t)gx + (uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz;
  int kind = NK_FLUID;
  if constexpr (GENERAL) {
    const uint32_t code = p.map[gi];
    kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
    if (!kind_is_wet(kind)) return;
  }
  const AxisOff ox = axis_off(gx, g.lat_nx, 1, g.wrap[0]);
  const AxisOff oy = axis_off(gy, g.lat_ny, g.arr_nx, g.wrap[1]);
  const AxisOff oz = (L::dim == 3) ? axis_off(gz, g.lat_nz, g.arr_nxy, g.wrap[2]) : AxisOff{0, 0};
  const size_t ds = g.dist_size;
  const uint32_t* nodes = p.nodes;
  uint32_t si = gi;
  if (nodes) {
    si = nodes[gi];
    if (si == INVALID_NODE) return;
  }
  R f[L::Q];
  static_for<0, L::Q>([&](auto I) {
    if constexpr (PROP == PROP_AA_ODD) {
      const int off = dir_offset<L, I>(ox, oy, oz, false);
      uint32_t sn = (uint32_t)((int)gi + off);
      if (nodes) sn = nodes[sn];
      f[I] = (sn != INVALID_NODE) ? (p.din + ds * (size_t)L::opp(I))[sn] : (R)0;
    } else {
      f[I] = (p.din + ds * (size_t)I)[si];
    }
  });
  R rho, v[3];
  if (p.cp.incompressible == SLF_DENSITY_ROUNDOFF) macro_roundoff<L, R>(f, rho, v);
  else macro_standard<L, R>(f, p.cp.incompressible != 0, rho, v);
  p.rho[gi] = rho;
  p.vx[gi] = v[0];
  p.vy[gi] = v[1];
  if constexpr (L::dim == 3) p.vz[gi] = v[2];
}

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------

template <class L, class R, int MODEL, int PROP>
static hipError_t launch_sweep4(bool general, const Geometry& g, const Physics& ph, const SweepArgs& a, int y0, int y1,
                                int z0, int z1, int block_x, hipStream_t s) {
  const SweepParams<L, R> p = make_params<L, R>(g, ph, a, y0, z0);
  dim3 block(block_x, 1, 1);
  dim3 grid((g.lat_nx - 2 + block_x - 1) / block_x, y1 - y0, L::dim == 3 ? z1 - z0 : 1);
  if (grid.y == 0 || grid.z == 0) return hipSuccess;
  if constexpr (MODEL == 0) {
    if (ph.incompressible == SLF_DENSITY_ROUNDOFF) {       // --minimize_roundoff (BGK only; checked at module creation)
      if (g.indirect) hipLaunchKernelGGL((sweep_kernel<L, R, MODEL, PROP, true, true, true>), grid, block, 0, s, p);
      else if (general) hipLaunchKernelGGL((sweep_kernel<L, R, MODEL, PROP, true, false, true>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((sweep_kernel<L, R, MODEL, PROP, false, false, true>), grid, block, 0, s, p);
      return hipGetLastError();
    }
    if (ph.regularized || ph.subgrid) {       // --regularized / --subgrid=les-smagorinsky (BGK only; checked at module creation)
      if (block.x > 512) {                    // these instantiations: at most 512 threads per workgroup (launch bounds)
        block.x = 512;
        grid.x = (g.lat_nx - 2 + 511) / 512;
      }
      if (g.indirect) hipLaunchKernelGGL((sweep_kernel<L, R, MODEL, PROP, true, true, false, true>), grid, block, 0, s, p);
      else if (general) hipLaunchKernelGGL((sweep_kernel<L, R, MODEL, PROP, true, false, false, true>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((sweep_kernel<L, R, MODEL, PROP, false, false, false, true>), grid, block, 0, s, p);
      return hipGetLastError();
    }
  }
  if (g.indirect && p.slot_gi) {       // one thread per slot (slot_sweep_kernel)
    SweepParams<L, R> q = p;
    q.y1 = y1;
    q.z1 = (L::dim == 3) ? z1 : 1;
    return launch_slot_sweep<L, R>(MODEL, PROP, g.bc_level, q, s);      // slf_slots.hip
  }
  if (g.indirect) hipLaunchKernelGGL((sweep_kernel<L, R, MODEL, PROP, true, true>), grid, block, 0, s, p);
  else if (general) hipLaunchKernelGGL((sweep_kernel<L, R, MODEL, PROP, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((sweep_kernel<L, R, MODEL, PROP, false>), grid, block, 0, s, p);
  return hipGetLastError();
}

template <class L, class R, int MODEL>
static hipError_t launch_sweep3(Prop prop, bool general, const Geometry& g, const Physics& ph, const SweepArgs& a,
                                int y0, int y1, int z0, int z1, int bx, hipStream_t s) {
  switch (prop) {
    case PROP_AB: return launch_sweep4<L, R, MODEL, PROP_AB>(general, g, ph, a, y0, y1, z0, z1, bx, s);
    case PROP_AA_EVEN: return launch_sweep4<L, R, MODEL, PROP_AA_EVEN>(general, g, ph, a, y0, y1, z0, z1, bx, s);
    default: return launch_sweep4<L, R, MODEL, PROP_AA_ODD>(general, g, ph, a, y0, y1, z0, z1, bx, s);
  }
}

template <class L, class R>
static hipError_t launch_sweep2(int model, Prop prop, bool general, const Geometry& g, const Physics& ph,
                                const SweepArgs& a, int y0, int y1, int z0, int z1, int bx, hipStream_t s) {
  if (model == 0) return launch_sweep3<L, R, 0>(prop, general, g, ph, a, y0, y1, z0, z1, bx, s);
  return launch_sweep3<L, R, 1>(prop, general, g, ph, a, y0, y1, z0, z1, bx, s);
}

#define SLF_DISPATCH_LR(sel, CALL)                                   \
  do {                                                               \
    if ((sel).lattice == 0) {                                        \
      if ((sel).precision == 4) { using L = D2Q9; using R = float; CALL; }  \
      else { using L = D2Q9; using R = double; CALL; }               \
    } else {                                                         \
      if ((sel).precision == 4) { using L = D3Q19; using R = float; CALL; } \
      else { using L = D3Q19; using R = double; CALL; }              \
    }                                                                \
  } while (0)

hipError_t launch_sweep(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                        const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x, hipStream_t s) {
  if (!g.indirect) {   // the tuned kernels address the dense layout
    hipError_t fe = hipSuccess;
    if (launch_sweep_fast(sel, prop, g, ph, a, y0, y1, z0, z1, block_x, s, &fe)) return fe;
    if (launch_sweep_row(sel, prop, g, ph, a, y0, y1, z0, z1, s, &fe)) return fe;
  }
  SLF_DISPATCH_LR(sel, return (launch_sweep2<L, R>(sel.model, prop, sel.general, g, ph, a, y0, y1, z0, z1, block_x, s)));
  return hipErrorInvalidValue;
}

template <class L, class R>
static hipError_t launch_init2(const Geometry& g, const Physics& ph, void* dist, const void* rho, const void* const v[3],
                               const void* nodes, hipStream_t s) {
  dim3 block(256, 1, 1);
  dim3 grid((g.lat_nx + 255) / 256, g.lat_ny, g.lat_nz);
  hipLaunchKernelGGL((init_kernel<L, R>), grid, block, 0, s, (R*)dist, (const R*)rho, (const R*)v[0], (const R*)v[1],
                     (const R*)v[2], g, ph.incompressible, (const uint32_t*)nodes);
  return hipGetLastError();
}

hipError_t launch_init(const KernelSelector& sel, const Geometry& g, const Physics& ph, void* dist, const void* rho,
                       const void* const v[3], const void* nodes, hipStream_t s) {
  SLF_DISPATCH_LR(sel, return (launch_init2<L, R>(g, ph, dist, rho, v, nodes, s)));
  return hipErrorInvalidValue;
}

template <class L, class R>
static hipError_t launch_pbc2(const Geometry& g, void* dist, int axis, bool with_swap, hipStream_t s) {
  const int lat[3] = {g.lat_nx, g.lat_ny, g.lat_nz};
  int b_ax = (axis == 0) ? 1 : 0;
  int c_ax = (axis == 2) ? 1 : 2;
  dim3 block(256, 1, 1);
  dim3 grid((lat[b_ax] + 255) / 256, L::dim == 3 ? lat[c_ax] : 1, 1);
  if (with_swap) hipLaunchKernelGGL((pbc_kernel<L, R, true>), grid, block, 0, s, (R*)dist, g, axis);
  else hipLaunchKernelGGL((pbc_kernel<L, R, false>), grid, block, 0, s, (R*)dist, g, axis);
  return hipGetLastError();
}

hipError_t launch_pbc(const KernelSelector& sel, const Geometry& g, void* dist, int axis, bool with_swap,
                      hipStream_t s) {
  if (axis < 0 || axis >= g.dim) return hipErrorInvalidValue;
  SLF_DISPATCH_LR(sel, return (launch_pbc2<L, R>(g, dist, axis, with_swap, s)));
  return hipErrorInvalidValue;
}

hipError_t launch_macro_pbc(const KernelSelector& sel, const Geometry& g, void* field, int axis, hipStream_t s) {
  if (axis < 0 || axis >= g.dim) return hipErrorInvalidValue;
  const int lat[3] = {g.lat_nx, g.lat_ny, g.lat_nz};
  int b_ax = (axis == 0) ? 1 : 0;
  int c_ax = (axis == 2) ? 1 : 2;
  dim3 block(256, 1, 1);
  dim3 grid((lat[b_ax] + 255) / 256, lat[c_ax], 1);
  if (sel.precision == 4) hipLaunchKernelGGL((macro_pbc_kernel<float>), grid, block, 0, s, (float*)field, g, axis);
  else hipLaunchKernelGGL((macro_pbc_kernel<double>), grid, block, 0, s, (double*)field, g, axis);
  return hipGetLastError();
}

hipError_t launch_sparse(const KernelSelector& sel, bool collect, const unsigned long long* idx, void* dist,
                         void* buffer, int n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  dim3 block(256, 1, 1);
  dim3 grid((n + 255) / 256, 1, 1);
  if (sel.precision == 4) {
    if (collect) hipLaunchKernelGGL((sparse_kernel<float, true>), grid, block, 0, s, idx, (float*)dist, (float*)buffer, n);
    else hipLaunchKernelGGL((sparse_kernel<float, false>), grid, block, 0, s, idx, (float*)dist, (float*)buffer, n);
  } else {
    if (collect) hipLaunchKernelGGL((sparse_kernel<double, true>), grid, block, 0, s, idx, (double*)dist, (double*)buffer, n);
    else hipLaunchKernelGGL((sparse_kernel<double, false>), grid, block, 0, s, idx, (double*)dist, (double*)buffer, n);
  }
  return hipGetLastError();
}

template <class L, class R>
static hipError_t launch_macro2(Prop prop, bool general, const Geometry& g, const Physics& ph, const SweepArgs& a,
                                hipStream_t s) {
  const SweepParams<L, R> p = make_params<L, R>(g, ph, a, 1, L::dim == 3 ? 1 : 0);
  const int bx = 256;
  dim3 block(bx, 1, 1);
  dim3 grid((g.lat_nx - 2 + bx - 1) / bx, g.lat_ny - 2, L::dim == 3 ? g.lat_nz - 2 : 1);
  if (prop == PROP_AA_ODD) {
    if (general) hipLaunchKernelGGL((macro_kernel<L, R, PROP_AA_ODD, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((macro_kernel<L, R, PROP_AA_ODD, false>), grid, block, 0, s, p);
  } else {
    if (general) hipLaunchKernelGGL((macro_kernel<L, R, PROP_AB, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((macro_kernel<L, R, PROP_AB, false>), grid, block, 0, s, p);
  }
  return hipGetLastError();
}

hipError_t launch_macro(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                        const SweepArgs& a, hipStream_t s) {
  SLF_DISPATCH_LR(sel, return (launch_macro2<L, R>(prop, sel.general, g, ph, a, s)));
  return hipErrorInvalidValue;
}

}  // namespace slf
