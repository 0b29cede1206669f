// Indirect addressing launched over the SLOTS of the active-node arrays (gfx950): slot_sweep_kernel, the slot -> node table.
// A translation unit of its own: 3 boundary-condition levels x 3 steps x 2 models x 2 lattices x 2 precisions of the full node
// code compile as long as everything else of slf_kernels.hip together, and the library's sources are compiled in parallel
// (sailfish_amd/build.py).
#include "../../include/sailfish_hip.h"
#include "slf_kernels.h"
#include "slf_node.h"
#include "slf_sweep.h"

namespace slf {

// Indirect addressing (reference subdomain_runner.py:829-878, kernel_common.mako:140-167), one thread per SLOT: lane l of
// a wave owns slot s0 + l of every array, so the own-slot accesses of the even in-place step are whole lines and every
// lane works -- the per-node launch above walks the dense box, where a packed bed (30 % fluid) leaves 70 % of the lanes
// of every wave idle and the active ones' accesses a third of a line (6.5 GMLUPS; profiles/r06/configs_indirect.jsonl).
// The x-streaming steps translate every neighbour ONCE: node x - e_i (the pull) is node x + e_opp(i) (the push), so the
// 18 entries of the dense table serve both; consecutive active nodes of a row have consecutive slots, so the gathers
// through them are as good as dense.  Same node code (node_update<..., INDIRECT>), same results.
// BCL: the module's boundary-condition level (Geometry::bc_level, as the whole-row kernels): a porous medium or a pipe of
// bounce-back walls is level 0 and carries none of the outflow / slip / do-nothing code, whose merged populations the
// level-2 instantiation of the two-copy step keeps in 32 bytes of scratch per lane at its 128 VGPRs.
// Double precision D3Q19: two waves per SIMD, i.e. 256 VGPRs -- at the 128 of four waves every instantiation kept 100-330
// bytes per lane in scratch, more than the 304 bytes of populations a node moves.
template <class L, class R, int MODEL, int PROP, int BCL = 2>
__global__ void __launch_bounds__(256, (sizeof(R) == 8 && L::Q > 9) ? 2 : 4) slot_sweep_kernel(const SweepParams<L, R> p) {
  const Geometry& g = p.g;
  const uint32_t si = blockIdx.x * 256u + threadIdx.x;
  if (si >= p.n_slots) return;
  const uint32_t gi = p.slot_gi[si];
  if (gi == INVALID_NODE) return;
  const uint32_t yz = p.slot_yz[si];
  const int gy = (int)(yz & 0xffffu), gz = (int)(yz >> 16);
  if (gy < p.y0 || gy >= p.y1) return;                       // launches over a region of the subdomain (boundary / bulk)
  if (L::dim == 3 && (gz < p.z0 || gz >= p.z1)) return;
  const int gx = (int)(gi - ((uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz));
  if (gx < 1 || gx > g.lat_nx - 2) return;                   // the layer of ghost nodes owns slots too
  const uint32_t code = p.map[gi];
  const int kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
  if (kind_is_excluded(kind)) return;
  const AxisOff ox = axis_off(gx, g.lat_nx, 1, g.wrap[0]);
  const AxisOff oy = axis_off(gy, g.lat_ny, g.arr_nx, g.wrap[1]);
  const AxisOff oz = (L::dim == 3) ? axis_off(gz, g.lat_nz, g.arr_nxy, g.wrap[2]) : AxisOff{0, 0};
  const size_t ds = g.dist_size;
  uint32_t t[L::Q];            // slot of the node at x + e_i
  t[0] = si;
  if constexpr (PROP != PROP_AA_EVEN) {
    static_for<1, L::Q>([&](auto I) { t[I] = p.nodes[(uint32_t)((int)gi + dir_offset<L, I>(ox, oy, oz, true))]; });
  }
  R f[L::Q];
  static_for<0, L::Q>([&](auto I) {
    if constexpr (PROP == PROP_AA_ODD) {
      const uint32_t sn = t[L::opp(I)];                      // x - e_i = x + e_opp(i)
      f[I] = (sn != INVALID_NODE) ? (p.din + ds * (size_t)L::opp(I))[sn] : (R)0;
    } else {
      f[I] = (p.din + ds * (size_t)I)[si];
    }
  });
  R rho, v[3];
  bool wet = true;
  node_update<L, R, MODEL, PROP, true, true, FORCE_RUNTIME, BCL, false>(p, f, code, kind, gi, ox, oy, oz, rho, v, wet, si);
  if (wet) check_invalid<R>(p.status, p.options, rho, gx, gy, gz);
  if ((p.options & 1u) && wet) {
    p.rho[gi] = rho;
    p.vx[gi] = v[0];
    p.vy[gi] = v[1];
    if constexpr (L::dim == 3) p.vz[gi] = v[2];
  }
  static_for<0, L::Q>([&](auto I) {
    if constexpr (PROP == PROP_AA_EVEN) {
      (p.dout + ds * (size_t)L::opp(I))[si] = f[I];
    } else {
      if (t[I] != INVALID_NODE) (p.dout + ds * (size_t)I)[t[I]] = f[I];
    }
  });
}

// slot -> node, from the dense node -> slot table: one thread per node of the padded box
__global__ void __launch_bounds__(256) build_slot_table_kernel(const uint32_t* __restrict__ nodes, Geometry g, uint32_t* slot_gi,
                                                               uint32_t* slot_yz, uint32_t* max_slot) {
  const int gx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int gy = (int)blockIdx.y, gz = (int)blockIdx.z;
  if (gx >= g.lat_nx) return;
  const uint32_t gi = (uint32_t)gx + (uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz;
  const uint32_t s = nodes[gi];
  if (s == INVALID_NODE || s >= g.dist_size) return;
  slot_gi[s] = gi;
  slot_yz[s] = (uint32_t)gy | ((uint32_t)gz << 16);
  atomicMax(max_slot, s);
}

hipError_t launch_build_slot_table(const Geometry& g, const void* nodes, uint32_t* slot_gi, uint32_t* slot_yz,
                                   uint32_t* max_slot, hipStream_t s) {
  dim3 block(256, 1, 1);
  dim3 grid((g.lat_nx + 255) / 256, g.lat_ny, g.lat_nz);
  hipLaunchKernelGGL(build_slot_table_kernel, grid, block, 0, s, (const uint32_t*)nodes, g, slot_gi, slot_yz, max_slot);
  return hipGetLastError();
}

template <class L, class R, int MODEL, int PROP>
static hipError_t slot_sweep3(int bc_level, const SweepParams<L, R>& q, hipStream_t s) {
  const dim3 sgrid((q.n_slots + 255) / 256, 1, 1), sblock(256, 1, 1);
  // (single-precision BGK, odd in-place step: the level-0 instantiation comes out of the register allocator with 28 bytes
  // of scratch at 128 VGPRs where level 1 has none at 124 -- profiles/r06/kernels_resources.txt -- so level 1 serves both)
  constexpr bool skip0 = sizeof(R) == 4 && MODEL == 0 && PROP == PROP_AA_ODD && L::Q > 9;
  if constexpr (!skip0) {
    if (bc_level == 0) {
      hipLaunchKernelGGL((slot_sweep_kernel<L, R, MODEL, PROP, 0>), sgrid, sblock, 0, s, q);
      return hipGetLastError();
    }
  }
  if (bc_level <= 1) hipLaunchKernelGGL((slot_sweep_kernel<L, R, MODEL, PROP, 1>), sgrid, sblock, 0, s, q);
  else hipLaunchKernelGGL((slot_sweep_kernel<L, R, MODEL, PROP, 2>), sgrid, sblock, 0, s, q);
  return hipGetLastError();
}

template <class L, class R, int MODEL>
static hipError_t slot_sweep2(int prop, int bc_level, const SweepParams<L, R>& q, hipStream_t s) {
  if (prop == PROP_AB) return slot_sweep3<L, R, MODEL, PROP_AB>(bc_level, q, s);
  if (prop == PROP_AA_EVEN) return slot_sweep3<L, R, MODEL, PROP_AA_EVEN>(bc_level, q, s);
  return slot_sweep3<L, R, MODEL, PROP_AA_ODD>(bc_level, q, s);
}

template <class L, class R>
hipError_t launch_slot_sweep(int model, int prop, int bc_level, const SweepParams<L, R>& q, hipStream_t s) {
  if (model == 0) return slot_sweep2<L, R, 0>(prop, bc_level, q, s);
  return slot_sweep2<L, R, 1>(prop, bc_level, q, s);
}

template hipError_t launch_slot_sweep<D2Q9, float>(int, int, int, const SweepParams<D2Q9, float>&, hipStream_t);
template hipError_t launch_slot_sweep<D2Q9, double>(int, int, int, const SweepParams<D2Q9, double>&, hipStream_t);
template hipError_t launch_slot_sweep<D3Q19, float>(int, int, int, const SweepParams<D3Q19, float>&, hipStream_t);
template hipError_t launch_slot_sweep<D3Q19, double>(int, int, int, const SweepParams<D3Q19, double>&, hipStream_t);

}  // namespace slf
