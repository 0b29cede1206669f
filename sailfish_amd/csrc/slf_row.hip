// Whole-row sweep kernels: the aligned-streaming scheme of the tuned north-star kernel
// (slf_fast.hip: fast_row_kernel) for every D3Q19 configuration -- f32 / f64, BGK / MRT, with or without
// the node map (walls, boundary conditions, unused nodes), x wrapped in-sweep or not.
//
//   * one workgroup = one (y, z) row (or, opt-in, an x-segment of it: row_block_x), thread t of workgroup b
//     owns node x = 1 + b * blockDim.x + t; the distribution arrays are allocated so that x = 1 starts a
//     128-byte line  =>  every global access of a wave is line aligned;
//   * the +-1 x shift of the push (AB and odd AA step) happens in registers: a DPP wave shift inside a
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
//
// Row classes (RowClasses, slf_kernels.h; node-map instantiations): a wave whose 64 nodes are all plain fluid
// (segment class 0, one scalar load) does not read the map -- its lanes are fluid nodes by construction (a second,
// straight-line collision body for such waves was tried: two inlined collisions in one kernel cost 16-30 VGPRs and
// spill, profiles/r03/row_classes_resources.txt); rows with
// boundary-condition nodes are left to a second launch of the instantiation for the module's bc_level over the
// list of those rows (launch_row(), slf_sweep.h), so that everything else -- 99.8 % of a lid-driven cavity -- runs
// the BCL = 0 instantiation with its small register budget.
constexpr int NT = 3;   // populations are streamed once per step: non-temporal loads and stores

// Minimum resident waves per SIMD asked of the compiler (second launch bound; it budgets VGPRs *and* SGPRs for it --
// 800 SGPRs per SIMD make 104 of them a 7-wave kernel however few VGPRs it has).  The f32 node-map instantiations fit
// 6 without spilling (77-80 VGPRs instead of 81-91); those for tables without boundary-condition nodes and modules
// without a body force fit 8 (BGK, 46-53 VGPRs); MRT with the pair-form moment transform (slf_node.h) needs 54-58 VGPRs, 70
// in the load-first odd-step instantiation: asked for 7, everything but that one reaches 8 and nothing spills (asked for
// 8, the load-first instantiation spills 32 bytes); checked with -Rpass-analysis=kernel-resource-usage
// (tools/resource_usage.py, profiles/r03/kernels_resources_final.txt).
#ifndef SLF_MRT_L0_WAVES
#define SLF_MRT_L0_WAVES 7
#endif
template <class R, int MODEL, bool GENERAL, bool FORCE, int BCL>
constexpr int row_min_waves() {
  if (sizeof(R) == 4 && !GENERAL && MODEL == 0 && !FORCE) return 8;   // 46 VGPRs; the bound keeps it under 96 SGPRs too
  if (sizeof(R) != 4 || !GENERAL) return 4;
  if (BCL == 0 && !FORCE) return MODEL == 0 ? 8 : SLF_MRT_L0_WAVES;
  return 6;
}

template <class L, class R, int MODEL, int PROP, bool GENERAL, bool FORCE, bool SPEC = false, int BCL = 2>
__global__ void __launch_bounds__(1024, (row_min_waves<R, MODEL, GENERAL, FORCE, BCL>())) row_kernel(const SweepParams<L, R> p) {
  static_assert(PROP == PROP_AB || PROP == PROP_AA_ODD, "the even AA step has no x shift");
  const Geometry& g = p.g;
  int ry, rz;
  if (!launch_row(p, ry, rz)) return;          // the whole workgroup: before any barrier
  const int gy = sgpr(ry);
  const int gz = (L::dim == 3) ? sgpr(rz) : 0;
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
  // class of this wave's 64 nodes: 0 = all plain fluid (no map access)
  int cls = 1;
  if constexpr (GENERAL) cls = sgpr(segment_class(p, gy, gz, (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)));

  // ---- load.  Every address = (wave-uniform row base of the direction, in SGPRs) + (this lane's byte offset in
  // the row): uniform_base(), slf_sweep.h.  Odd AA step: pull with plain x-shifted loads -- misaligned *reads*
  // cost little (the neighbouring wave uses the rest of the line; measured equal to an LDS exchange of aligned
  // loads, profiles/r01/row_probe4.log) and save a barrier; the push below is what must be aligned.
  R f[L::Q];
  // Odd AA step next to a CONNECTED x face (fluid-only instantiations): what the edge node would pull out of the ghost
  // column arrives through the face buffer instead (x_face_receive overrides it), so its lane pulls from its own x -- a
  // line the wave fetches anyway -- rather than from the ghost column, which sits in a line of its own behind the row:
  // ten lines per row fetched from HBM for values nothing uses, +13 % reads on the 128-node rows of an eight-way x
  // split (profiles/r03/pmc_summary_cfg3*.txt).  Branch-free on purpose: predicating the loads instead made every wave
  // wait for all of its loads before the edge lanes' block (s_waitcnt vmcnt(0) at the join), 10 % slower on 1024-node rows.
  // Only where every entry the row reads is sure to have been written: fluid-only instantiations (with a node map the
  // sender's edge node may be excluded) and rows whose y and z neighbours are real or wrapped rows (the rows next to a
  // non-periodic y / z face pull from ghost rows, which no sweep of the neighbour has sent: those entries stay unset and
  // x_face_receive leaves what was pulled from the arrays -- the reference's behaviour, pinned by the propagation KATs).
  // The same select serves ghost columns that nothing stores into (slf_module_set_x_ghost_unused: the face is a wall or
  // open and its first real column holds no wet node): what a dry node pulls out of them only ever travels back into
  // them, and each of the ten x-moving directions fetched a line of its own for it -- 10 lines per row, +3.3 % reads on
  // a 512^3 cavity (profiles/r03/pmc_summary_cfg2b.txt).
  constexpr bool SKIP_GHOST_PULL = PROP == PROP_AA_ODD && !GENERAL;
  AxisOff oxl = ox;
  if constexpr (PROP == PROP_AA_ODD) {
    if (!g.wrap[0]) {
      if ((g.x_ghost_unused & 1) && x == 1) oxl.m = 0;
      if ((g.x_ghost_unused & 2) && x == nx) oxl.p = 0;
    }
  }
  if constexpr (SKIP_GHOST_PULL) {
    const bool inner = (g.wrap[1] || (gy > 1 && gy < g.lat_ny - 2)) &&
                       (L::dim < 3 || g.wrap[2] || (gz > 1 && gz < g.lat_nz - 2));
    if (inner) {
      if (p.xrecv[0] && x == 1) oxl.m = 0;
      if (p.xrecv[1] && x == nx) oxl.p = 0;
    }
  }
  auto src_of = [&](auto I) -> const SLF_GLOBAL R* {
    if constexpr (PROP == PROP_AA_ODD) {
      const int off = dir_offset<L, I>(ox0, oy, oz, false);                     // y, z part: uniform
      constexpr int ex = L::ex(I);
      const int xs = (ex == 0) ? 0 : ((ex > 0) ? oxl.m : oxl.p);                // x part of x - e_i: per lane
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
    if (cls != 0) {
      code = p.map[gi];
      kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
      active = live && !kind_is_excluded(kind);
    }
  }
  if constexpr (GENERAL && !SPEC) {
    static_for<0, L::Q>([&](auto I) { f[I] = active ? load(I) : (R)0; });
  }

  const FaceRows fr = face_rows<L>(g, gy, gz);
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
template <class L, class R, int MODEL, bool GENERAL, bool FORCE, bool SPEC = false, int BCL = 2>
__global__ void __launch_bounds__(1024, (BCL == 0 ? row_min_waves<R, MODEL, GENERAL, FORCE, BCL>() : ((GENERAL && sizeof(R) == 4 && MODEL == 1) ? 6 : 4))) even_kernel(const SweepParams<L, R> p) {
  const Geometry& g = p.g;
  int ry, rz;
  if (!launch_row(p, ry, rz)) return;
  const int gy = sgpr(ry);
  const int gz = (L::dim == 3) ? sgpr(rz) : 0;
  const int gx = 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  int cls = 1;
  if constexpr (GENERAL) cls = sgpr(segment_class(p, gy, gz, (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)));
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
    if (cls != 0) {
      code = p.map[gi];
      kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
      if (kind_is_excluded(kind)) return;
    }
  }
  const AxisOff ox = axis_off(gx, g.lat_nx, 1, g.wrap[0]);
  const AxisOff oy = axis_off(gy, g.lat_ny, g.arr_nx, g.wrap[1]);
  const AxisOff oz = (L::dim == 3) ? axis_off(gz, g.lat_nz, g.arr_nxy, g.wrap[2]) : AxisOff{0, 0};
  if constexpr (!(GENERAL && SPEC)) {
    static_for<0, L::Q>([&](auto I) { f[I] = ldg<NT>(at_byte(uniform_base(p.din + ds * (size_t)I + row), xb)); });
  }
  const FaceRows fr = face_rows<L>(g, gy, gz);
  x_face_receive<L, R, false>(p, f, gx, g.lat_nx - 2, fr);
  R rho, v[3];
  bool wet = true;
  node_update<L, R, MODEL, PROP_AA_EVEN, GENERAL, false, FORCE, BCL>(p, f, code, kind, gi, ox, oy, oz, rho, v, wet);
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

// One launch of the instantiation for boundary-condition level BCL.  variant bit 64 = the module descriptor says
// "sparse geometry" (many excluded nodes): predicate the loads on the node map instead of issuing them first.
template <class L, class R, int MODEL, bool GENERAL, bool FORCE, int BCL>
static void launch_level(Prop prop, const SweepParams<L, R>& p, dim3 grid, dim3 block, hipStream_t s) {
  const bool spec = GENERAL && !(p.g.variant & 64);
  switch (prop) {
    case PROP_AB:   // (no gain from SPEC here: row_probe8.log)
      hipLaunchKernelGGL((row_kernel<L, R, MODEL, PROP_AB, GENERAL, FORCE, false, BCL>), grid, block, 0, s, p);
      break;
    case PROP_AA_ODD:
      if (spec) hipLaunchKernelGGL((row_kernel<L, R, MODEL, PROP_AA_ODD, GENERAL, FORCE, GENERAL, BCL>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((row_kernel<L, R, MODEL, PROP_AA_ODD, GENERAL, FORCE, false, BCL>), grid, block, 0, s, p);
      break;
    default:
      if (spec) hipLaunchKernelGGL((even_kernel<L, R, MODEL, GENERAL, FORCE, GENERAL, BCL>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((even_kernel<L, R, MODEL, GENERAL, FORCE, false, BCL>), grid, block, 0, s, p);
      break;
  }
}

template <class L, class R, int MODEL, bool GENERAL, bool FORCE>
static void launch_row5(Prop prop, SweepParams<L, R> p, int nx, int ny, int nz, const RowClasses* rc, hipStream_t s) {
  const int bx = row_block_x(nx);
  dim3 block(bx, 1, 1);
  dim3 grid((nx + bx - 1) / bx, ny, nz);
  p.y1 = p.y0 + ny;
  p.z1 = p.z0 + nz;
  // connected x faces: the rows that share the lines of the face buffers go to one XCD (xcd_row(), slf_sweep.h)
  const int xs = (p.xsend[0] || p.xsend[1] || p.xrecv[0] || p.xrecv[1]) ? xcd_shift_for((unsigned)ny, grid.x) : 0;
  p.row_mode = xs << 4;
  // node-map kernels: one instantiation per Geometry::bc_level (slf_kernels.h).  (Double precision too: the 64 bytes
  // of scratch per lane of its two-copy kernels are the outflow-node code of level 2, not the 128-VGPR cap of a
  // 1024-thread workgroup -- they stay with 512-thread workgroups: profiles/r03/f64_row_kernels_resources.txt)
  if constexpr (GENERAL) {
    const int bcl = p.g.bc_level;
    if (bcl == 0) {
      launch_level<L, R, MODEL, GENERAL, FORCE, 0>(prop, p, grid, block, s);
      return;
    }
    if (rc && p.seg_class) {
      // rows without boundary-condition nodes: level 0; the others (listed): the module's level
      if (rc->n_bc_rows < rc->n_rows) {
        p.row_mode = (rc->n_bc_rows ? 1 : 0) | (xs << 4);
        launch_level<L, R, MODEL, GENERAL, FORCE, 0>(prop, p, grid, block, s);
      }
      if (rc->n_bc_rows) {
        p.row_mode = 2;
        grid = dim3(grid.x, rc->n_bc_rows, 1);
        if (bcl == 1) launch_level<L, R, MODEL, GENERAL, FORCE, 1>(prop, p, grid, block, s);
        else launch_level<L, R, MODEL, GENERAL, FORCE, 2>(prop, p, grid, block, s);
      }
      return;
    }
    if (bcl == 1) {
      launch_level<L, R, MODEL, GENERAL, FORCE, 1>(prop, p, grid, block, s);
      return;
    }
  }
  launch_level<L, R, MODEL, GENERAL, FORCE, 2>(prop, p, grid, block, s);
}

template <class L, class R, int MODEL, bool GENERAL>
static void launch_row4(Prop prop, const SweepParams<L, R>& p, int nx, int ny, int nz, const RowClasses* rc, hipStream_t s) {
  // compile-time "body force or not" instantiations: see bgk_relax, slf_node.h
  if (p.cp.has_force != 0) launch_row5<L, R, MODEL, GENERAL, true>(prop, p, nx, ny, nz, rc, s);
  else launch_row5<L, R, MODEL, GENERAL, false>(prop, p, nx, ny, nz, rc, s);
}

template <class L, class R>
static void launch_row2(const KernelSelector& sel, Prop prop, const SweepParams<L, R>& p, int nx, int ny, int nz,
                        const RowClasses* rc, hipStream_t s) {
  if (sel.model == 0) {
    if (sel.general) launch_row4<L, R, 0, true>(prop, p, nx, ny, nz, rc, s);
    else launch_row4<L, R, 0, false>(prop, p, nx, ny, nz, rc, s);
  } else {
    if (sel.general) launch_row4<L, R, 1, true>(prop, p, nx, ny, nz, rc, s);
    else launch_row4<L, R, 1, false>(prop, p, nx, ny, nz, rc, s);
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
  const RowClasses* rc = (sel.general && a.rows && a.rows->map == a.map) ? a.rows : nullptr;
  if (sel.precision == 4) launch_row2<D3Q19, float>(sel, prop, make_params<D3Q19, float>(g, ph, a, y0, z0), nx, ny, nz, rc, s);
  else launch_row2<D3Q19, double>(sel, prop, make_params<D3Q19, double>(g, ph, a, y0, z0), nx, ny, nz, rc, s);
  *err = hipGetLastError();
  return true;
}

// ---- row classification (slf_module_classify_rows) ------------------------------------------------------------
// One workgroup per real (y, z) row; wave w looks at the segments w, w + 4, ...: class of a segment = the worst
// node kind among its live lanes (0 plain fluid, 1 no boundary-condition code, 2 boundary conditions).
__global__ void __launch_bounds__(256) classify_rows_kernel(const uint32_t* __restrict__ map, Geometry g, uint8_t* seg_class,
                                                            uint8_t* row_class, uint32_t* bc_rows, uint32_t* counters,
                                                            int nseg) {
  __shared__ int s_max, s_fluid;
  const int gy = 1 + (int)blockIdx.y;
  const int gz = (g.dim == 3) ? 1 + (int)blockIdx.z : 0;
  const int nx = g.lat_nx - 2;
  const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6;
  if (threadIdx.x == 0) { s_max = 0; s_fluid = 0; }
  __syncthreads();
  const uint32_t rowidx = (uint32_t)(gy + g.arr_ny * gz);
  const uint32_t row = (uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz;
  int wmax = 0, wfluid = 0;
  for (int sgm = w; sgm < nseg; sgm += 4) {
    const int x = 1 + sgm * 64 + lane;
    int level = 0;
    if (x <= nx) {
      const uint32_t code = map[row + (uint32_t)x];
      const int kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
      const bool plain = kind_is_excluded(kind) || kind == NK_FULL_BB;
      level = (kind == NK_FLUID) ? 0 : (plain ? 1 : 2);
    }
    const int c = __any(level == 2) ? 2 : (__any(level == 1) ? 1 : 0);
    if (lane == 0) seg_class[rowidx * (uint32_t)nseg + (uint32_t)sgm] = (uint8_t)c;
    wmax = c > wmax ? c : wmax;
    wfluid += c == 0 ? 1 : 0;
  }
  if (lane == 0) {
    atomicMax(&s_max, wmax);
    atomicAdd(&s_fluid, wfluid);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    row_class[rowidx] = (uint8_t)s_max;
    if (s_max == 2) bc_rows[atomicAdd(&counters[0], 1u)] = (uint32_t)gy | ((uint32_t)gz << 16);
    atomicAdd(&counters[1], (uint32_t)s_fluid);
  }
}

hipError_t launch_classify_rows(const Geometry& g, const void* map, uint32_t* seg_class, uint8_t* row_class,
                                uint32_t* bc_rows, uint32_t* counters, int nseg, hipStream_t s) {
  dim3 block(256, 1, 1);
  dim3 grid(1, g.lat_ny - 2, g.dim == 3 ? g.lat_nz - 2 : 1);
  hipLaunchKernelGGL(classify_rows_kernel, grid, block, 0, s, (const uint32_t*)map, g, (uint8_t*)seg_class, row_class,
                     bc_rows, counters, nseg);
  return hipGetLastError();
}

}  // namespace slf
