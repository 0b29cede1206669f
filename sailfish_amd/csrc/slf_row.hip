// Whole-row sweep kernels: the aligned-streaming scheme of the tuned north-star kernel
// (slf_fast.hip: fast_row_kernel) for every D3Q19 configuration -- f32 / f64, BGK / MRT, with or without
// the node map (walls, boundary conditions, unused nodes), x wrapped in-sweep or not.
//
//   * one workgroup = one (y, z) row (or, opt-in, an x-segment of it: row_block_x), thread t of workgroup b
//     owns node x = 1 + b * blockDim.x + t; the distribution arrays are allocated so that x = 1 starts a
//     128-byte line  =>  every global access of a wave is line aligned;
//   * the +-1 x shift of the push (AB and odd AA step) happens in registers: __shfl_up/down inside a
//     wave64, one LDS word per direction between neighbouring waves, the periodic wrap as the cyclic
//     continuation (x = nx <-> x = 1)  =>  every *store* is aligned (partial-line writes are what hurts
//     HBM); the pull of the odd AA step uses plain shifted loads (misaligned reads are served from cache);
//   * without in-sweep wrap the ghost columns x = 0 and x = nx + 1 are written by the two edge lanes of
//     the row only;
//   * a population is stored by the thread that owns its *target* x, if and only if its *source* node
//     takes part in the sweep (the reference pushes from every non-excluded node, whatever the target
//     is: propagation.mako:384-421) -- the 'active' flag travels with the value;
//   * populations are streamed once per step: non-temporal loads and stores.
//
// Arithmetic is node_update() of slf_sweep.h, shared with the per-node kernel in slf_kernels.hip: the
// results are bit-identical.  Replaces the reference's shared-memory propagation
// (templates/propagation.mako:180-288) on MI355X.
#include "slf_rowpush.h"

namespace slf {

// SPEC (node-map kernels): issue the population loads BEFORE the node map is read instead of predicating
// them on it.  A workgroup advances at the pace of its slowest wave (barrier in row_push), so the extra
// dependent round trip map -> loads costs 4-6 % of the odd step at 512^3 (profiles/r01/row_probe7.log,
// row_probe8.log); loading for excluded nodes as well wastes their bytes, so the host asks for SPEC only when
// few nodes are excluded.
template <class L, class R, int MODEL, int PROP, bool GENERAL, int NT, bool FORCE, bool SPEC = false, int BCL = 2>
// (second launch bound = minimum resident waves per SIMD: the f32 node-map instantiations fit 6 without spilling --
// 77-80 VGPRs instead of 81-91 -- which is worth a wave of occupancy; checked with -Rpass-analysis=kernel-resource-usage.
// Not the BGK instantiations with a body force: their Guo / exact-difference branches need ~94 and would spill 60-100 B)
__global__ void __launch_bounds__(1024, (GENERAL && sizeof(R) == 4) ? 6 : 4) row_kernel(const SweepParams<L, R> p) {
  static_assert(PROP == PROP_AB || PROP == PROP_AA_ODD, "the even AA step has no x shift");
  const Geometry& g = p.g;
  const int gy = sgpr(p.y0 + (int)blockIdx.y);
  const int gz = (L::dim == 3) ? sgpr(p.z0 + (int)blockIdx.z) : 0;
  const int nx = g.lat_nx - 2;
  const int x = 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const bool live = x <= nx;
  const uint32_t row = sgpr((uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz);
  const uint32_t xi = (uint32_t)(live ? x : 1);  // idle lanes: an in-row address, never stored
  const uint32_t gi = row + xi;
  const AxisOff ox = axis_off(x, g.lat_nx, 1, g.wrap[0]);
  AxisOff oy = axis_off(gy, g.lat_ny, g.arr_nx, g.wrap[1]);
  AxisOff oz = (L::dim == 3) ? axis_off(gz, g.lat_nz, g.arr_nxy, g.wrap[2]) : AxisOff{0, 0};
  oy.p = sgpr(oy.p); oy.m = sgpr(oy.m); oz.p = sgpr(oz.p); oz.m = sgpr(oz.m);
  const AxisOff ox0 = {0, 0};
  const size_t ds = g.dist_size;

  // ---- load.  Every address = (wave-uniform row base of the direction, in SGPRs) + (this lane's byte offset in
  // the row): uniform_base(), slf_sweep.h.  Odd AA step: pull with plain x-shifted loads -- misaligned *reads*
  // cost little (the neighbouring wave uses the rest of the line; measured equal to an LDS exchange of aligned
  // loads, profiles/r01/row_probe4.log) and save a barrier; the push below is what must be aligned.
  R f[L::Q];
  auto src_of = [&](auto I) -> const SLF_GLOBAL R* {
    if constexpr (PROP == PROP_AA_ODD) {
      const int off = dir_offset<L, I>(ox0, oy, oz, false);                     // y, z part: uniform
      constexpr int ex = L::ex(I);
      const int xs = (ex == 0) ? 0 : ((ex > 0) ? ox.m : ox.p);                  // x part of x - e_i: per lane
      return at_byte(uniform_base(p.din + ds * (size_t)L::opp(I) + (uint32_t)((int)row + off)),
                     (uint32_t)((int)xi + xs) * (uint32_t)sizeof(R));
    } else {
      return at_byte(uniform_base(p.din + ds * (size_t)I + row), xi * (uint32_t)sizeof(R));
    }
  };
  // (the x-shifted pulls of the odd step share every line with the neighbouring wave: no non-temporal hint on those)
  auto load = [&](auto I) -> R {
    constexpr int nt = (PROP == PROP_AA_ODD && L::ex(I) != 0) ? (NT & ~1) : NT;
    return ldg<nt>(src_of(I));
  };
  if constexpr (SPEC || !GENERAL) {
    static_for<0, L::Q>([&](auto I) { f[I] = load(I); });
  }
  int kind = NK_FLUID;
  uint32_t code = 0;
  bool active = live;
  if constexpr (GENERAL) {
    code = p.map[gi];
    kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
    active = live && !kind_is_excluded(kind);
  }
  if constexpr (GENERAL && !SPEC) {
    static_for<0, L::Q>([&](auto I) { f[I] = active ? load(I) : (R)0; });
  }

  const FaceRows fr = face_rows(g, gy, gz);
  if (active) x_face_receive<L, R, PROP == PROP_AA_ODD>(p, f, x, nx, fr);
  R rho, v[3];
  bool wet = true;
  if (active) {
    node_update<L, R, MODEL, PROP, GENERAL, false, FORCE, BCL>(p, f, code, kind, gi, ox, oy, oz, rho, v, wet);
    if (wet) check_invalid<R>(p.status, p.options, rho, x, gy, gz);
    if ((p.options & 1u) && wet) {
      p.rho[gi] = rho;
      p.vx[gi] = v[0];
      p.vy[gi] = v[1];
      if constexpr (L::dim == 3) p.vz[gi] = v[2];
    }
  }

  row_push<L, R, GENERAL, NT>(g, f, p.dout, ds, row, xi, x, nx, live, active, oy, oz, p.xsend, &fr);
}

// Even AA step: every access is to the node's own slots (aligned); per-node kernel with cache hints.
// SPEC (node-map instantiations, dense geometries): issue the 19 loads before the node map has arrived, as in
// row_kernel -- the map -> loads dependency is the longest chain of this kernel; excluded nodes then load in vain.
template <class L, class R, int MODEL, bool GENERAL, int NT, bool FORCE, bool SPEC = false>
__global__ void __launch_bounds__(1024, (GENERAL && sizeof(R) == 4 && MODEL == 1) ? 6 : 4) even_kernel(const SweepParams<L, R> p) {
  const Geometry& g = p.g;
  const int gy = sgpr(p.y0 + (int)blockIdx.y);
  const int gz = (L::dim == 3) ? sgpr(p.z0 + (int)blockIdx.z) : 0;
  const int gx = 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (gx > g.lat_nx - 2) return;
  const uint32_t row = sgpr((uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz);
  const uint32_t gi = row + (uint32_t)gx;
  const uint32_t xb = (uint32_t)gx * (uint32_t)sizeof(R);
  const size_t ds = g.dist_size;
  R f[L::Q];
  int kind = NK_FLUID;
  uint32_t code = 0;
  if constexpr (GENERAL && SPEC) {
    static_for<0, L::Q>([&](auto I) { f[I] = ldg<NT>(at_byte(uniform_base(p.din + ds * (size_t)I + row), xb)); });
  }
  if constexpr (GENERAL) {
    code = p.map[gi];
    kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
    if (kind_is_excluded(kind)) return;
  }
  const AxisOff ox = axis_off(gx, g.lat_nx, 1, g.wrap[0]);
  const AxisOff oy = axis_off(gy, g.lat_ny, g.arr_nx, g.wrap[1]);
  const AxisOff oz = (L::dim == 3) ? axis_off(gz, g.lat_nz, g.arr_nxy, g.wrap[2]) : AxisOff{0, 0};
  if constexpr (!(GENERAL && SPEC)) {
    static_for<0, L::Q>([&](auto I) { f[I] = ldg<NT>(at_byte(uniform_base(p.din + ds * (size_t)I + row), xb)); });
  }
  const FaceRows fr = face_rows(g, gy, gz);
  x_face_receive<L, R, false>(p, f, gx, g.lat_nx - 2, fr);
  R rho, v[3];
  bool wet = true;
  node_update<L, R, MODEL, PROP_AA_EVEN, GENERAL, false, FORCE>(p, f, code, kind, gi, ox, oy, oz, rho, v, wet);
  if (wet) check_invalid<R>(p.status, p.options, rho, gx, gy, gz);
  if ((p.options & 1u) && wet) {
    p.rho[gi] = rho;
    p.vx[gi] = v[0];
    p.vy[gi] = v[1];
    if constexpr (L::dim == 3) p.vz[gi] = v[2];
  }
  static_for<0, L::Q>([&](auto I) { stg<NT>(at_byte(uniform_base(p.dout + ds * (size_t)L::opp(I) + row), xb), f[I]); });
  x_face_send_own_row<L, R>(p, f, gx, g.lat_nx - 2, fr);
}

template <class L, class R, int MODEL, bool GENERAL, int NT, bool FORCE>
static void launch_row5(Prop prop, const SweepParams<L, R>& p, int nx, int ny, int nz, hipStream_t s) {
  const int bx = row_block_x(nx, p.g.variant);
  dim3 block(bx, 1, 1);
  dim3 grid((nx + bx - 1) / bx, ny, nz);
  // node-map kernels in single precision: one instantiation per Geometry::bc_level (slf_kernels.h)
  constexpr bool LEVELS = GENERAL && sizeof(R) == 4;
  const int bcl = LEVELS ? p.g.bc_level : 2;
  switch (prop) {
    case PROP_AB:   // (no gain from SPEC here: row_probe8.log)
      if constexpr (LEVELS) {
        if (bcl == 0) { hipLaunchKernelGGL((row_kernel<L, R, MODEL, PROP_AB, GENERAL, NT, FORCE, false, 0>), grid, block, 0, s, p); break; }
        if (bcl == 1) { hipLaunchKernelGGL((row_kernel<L, R, MODEL, PROP_AB, GENERAL, NT, FORCE, false, 1>), grid, block, 0, s, p); break; }
      }
      hipLaunchKernelGGL((row_kernel<L, R, MODEL, PROP_AB, GENERAL, NT, FORCE>), grid, block, 0, s, p);
      break;
    case PROP_AA_ODD:
      // variant bit 64 = the module descriptor says "sparse geometry" (many excluded nodes): predicate
      if constexpr (LEVELS) {
        if (bcl == 0) {
          if (!(p.g.variant & 64)) hipLaunchKernelGGL((row_kernel<L, R, MODEL, PROP_AA_ODD, GENERAL, NT, FORCE, true, 0>), grid, block, 0, s, p);
          else hipLaunchKernelGGL((row_kernel<L, R, MODEL, PROP_AA_ODD, GENERAL, NT, FORCE, false, 0>), grid, block, 0, s, p);
          break;
        }
      }
      if (GENERAL && !(p.g.variant & 64)) hipLaunchKernelGGL((row_kernel<L, R, MODEL, PROP_AA_ODD, GENERAL, NT, FORCE, GENERAL>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((row_kernel<L, R, MODEL, PROP_AA_ODD, GENERAL, NT, FORCE>), grid, block, 0, s, p);
      break;
    default:
      if (GENERAL && !(p.g.variant & 64)) hipLaunchKernelGGL((even_kernel<L, R, MODEL, GENERAL, NT, FORCE, GENERAL>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((even_kernel<L, R, MODEL, GENERAL, NT, FORCE>), grid, block, 0, s, p);
      break;
  }
}

template <class L, class R, int MODEL, bool GENERAL>
static void launch_row4(Prop prop, int nt, const SweepParams<L, R>& p, int nx, int ny, int nz, hipStream_t s) {
  const bool force = p.cp.has_force != 0;   // compile-time "no body force" instantiations: see bgk_relax, slf_node.h
  if (nt == 3) {
    if (force) launch_row5<L, R, MODEL, GENERAL, 3, true>(prop, p, nx, ny, nz, s);
    else launch_row5<L, R, MODEL, GENERAL, 3, false>(prop, p, nx, ny, nz, s);
  } else {
    if (force) launch_row5<L, R, MODEL, GENERAL, 0, true>(prop, p, nx, ny, nz, s);
    else launch_row5<L, R, MODEL, GENERAL, 0, false>(prop, p, nx, ny, nz, s);
  }
}

template <class L, class R>
static void launch_row2(const KernelSelector& sel, Prop prop, int nt, const SweepParams<L, R>& p, int nx, int ny,
                        int nz, hipStream_t s) {
  if (sel.model == 0) {
    if (sel.general) launch_row4<L, R, 0, true>(prop, nt, p, nx, ny, nz, s);
    else launch_row4<L, R, 0, false>(prop, nt, p, nx, ny, nz, s);
  } else {
    if (sel.general) launch_row4<L, R, 1, true>(prop, nt, p, nx, ny, nz, s);
    else launch_row4<L, R, 1, false>(prop, nt, p, nx, ny, nz, s);
  }
}

bool launch_sweep_row(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph, const SweepArgs& a,
                      int y0, int y1, int z0, int z1, hipStream_t s, hipError_t* err) {
  const int nx = g.lat_nx - 2;
  // 3-D only: in 2-D (D2Q9, a few thousand rows, ~20 us per sweep) the per-node kernel with its smaller
  // workgroups is 7-15 % faster (profiles/r01/row_general2.log)
  if (!(g.variant & 8) || sel.lattice != 1 || nx < 1) return false;
  const int ny = y1 - y0, nz = (g.dim == 3) ? z1 - z0 : 1;
  if (ny <= 0 || nz <= 0) return false;
  const int nt = (g.variant & 1) ? 3 : 0;
  if (sel.precision == 4) launch_row2<D3Q19, float>(sel, prop, nt, make_params<D3Q19, float>(g, ph, a, y0, z0), nx, ny, nz, s);
  else launch_row2<D3Q19, double>(sel, prop, nt, make_params<D3Q19, double>(g, ph, a, y0, z0), nx, ny, nz, s);
  *err = hipGetLastError();
  return true;
}

}  // namespace slf
