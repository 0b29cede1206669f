// Shared between slf_kernels.hip (general kernels) and slf_fast.hip (the tuned
// D3Q19 / f32 / fluid-only north-star kernels): parameter block and neighbour
// offset helpers of the sweep.
#pragma once
#include "slf_kernels.h"
#include "slf_node.h"

namespace slf {

template <class L, class R>
struct SweepParams {
  const uint32_t* __restrict__ nodes;   // indirect addressing table or NULL
  const uint32_t* __restrict__ map;
  const R* din;
  R* dout;
  R* rho;
  R* vx;
  R* vy;
  R* vz;
  const R* __restrict__ node_params;
  uint32_t* status;     // module-level {flag, x, y, z} for the on-GPU invalid value check (options bit 2)
  uint32_t options;
  int y0, z0;
  int relaxation_enabled;
  // x faces connected to another subdomain (1-D decompositions along x): what crosses the face is exchanged through
  // dense buffers [k][z][y] (k = position of the direction among those with e_x > 0 resp. < 0, rows over the whole
  // padded (arr_ny x arr_nz) plane) which the two edge lanes of a row write / read directly -- no ghost-column
  // stores in the sweep, no strided pack / unpack kernels (see x_face_* below).  NULL = face not connected.
  R* xsend[2];
  const R* xrecv[2];
  Geometry g;
  CollideParams<L, R> cp;
  // (new members go to the END: the kernel-argument layout decides how the scalar loads group, and the headline
  // kernels sit right at a VGPR boundary -- inserting these before `g` cost fast_even_kernel 12-20 bytes of scratch)
  // row classes of the node map (RowClasses, slf_kernels.h); seg_class == NULL: not classified, every wave reads the map.
  // row_mode 0: every row of the launch; 1: rows of class 2 are left to the list launch; 2: blockIdx.y indexes
  // bc_rows, rows outside [y0, y1) x [z0, z1) are skipped.
  const uint32_t* seg_class;
  const uint8_t* row_class;
  const uint32_t* bc_rows;
  int nseg, row_mode, y1, z1;
  // indirect addressing, launches over the slots (SlotTable, slf_kernels.h)
  const uint32_t* slot_gi;
  const uint32_t* slot_yz;
  uint32_t n_slots;
  // --regularized (bit 0) / --subgrid=les-smagorinsky (bit 1): read by the TURB instantiations of the per-node kernels only
  int turb_flags;
  R turb_visc, turb_c2x36;      // viscosity and 36 C^2 of the subgrid model
};


// Which rows an XCD gets.  Workgroups are dealt out to the eight XCDs round-robin by their flat index; with one
// workgroup per row every XCD sweeps every eighth row of a plane.  Where neighbouring rows share data through the L2 --
// the rho / phi rows a Shan-Chen force stencil reads (three rows per row, each last touched by another XCD), the words
// of a line of an x-face buffer (32 consecutive rows write one line, every L2 holds a piece of it) -- the rows of a block
// of 8 << s consecutive rows are handed out so that every XCD gets (1 << s) CONSECUTIVE ones.  All XCDs stay inside the
// same block of rows, and rows with the same y in neighbouring planes stay on the same XCD (launches whose row count is
// a multiple of the block; the hosts' xcd_shift_for() picks s accordingly, 0 = rows as they come).
__device__ __forceinline__ int xcd_row(int by, int s) {
  if (s == 0) return by;
  const int r = by & ((8 << s) - 1);
  return by - r + ((r & 7) << s) + (r >> 3);
}
static inline int xcd_shift_for(unsigned rows, unsigned grid_x) {
  static const int max_shift = [] {
    const char* e = getenv("SLF_XCD_ROWS_LOG2");      // rows per XCD and block = 1 << this; 0: off
    const int v = e ? atoi(e) : 5;
    return v < 0 ? 0 : (v > 8 ? 8 : v);
  }();
  if (grid_x != 1) return 0;
  int s = max_shift;
  while (s > 0 && (rows % (8u << s)) != 0) s--;
  return s;
}

// The (y, z) row a workgroup of a whole-row launch works on; false: nothing to do here (wave-uniform).
// row_mode: bits 0-3 the mode (below), bits 4+ the XCD row shift of xcd_row().
template <class L, class R>
__device__ __forceinline__ bool launch_row(const SweepParams<L, R>& p, int& gy, int& gz) {
  const int mode = p.row_mode & 15;
  if (mode == 2) {
    const uint32_t yz = p.bc_rows[blockIdx.y];
    gy = (int)(yz & 0xffffu);
    gz = (int)(yz >> 16);
    if (gy < p.y0 || gy >= p.y1 || (L::dim == 3 && (gz < p.z0 || gz >= p.z1))) return false;
  } else {
    gy = p.y0 + xcd_row((int)blockIdx.y, p.row_mode >> 4);
    gz = (L::dim == 3) ? p.z0 + (int)blockIdx.z : 0;
    if (mode == 1 && p.row_class[gy + p.g.arr_ny * gz] >= 2) return false;
  }
  return true;
}

// Class of the 64-node segment `seg` of row (gy, gz) through the scalar cache (the table is written once, by the
// classification kernel of an earlier launch): 0 = all plain fluid.  Without a table every segment counts as mixed.
template <class L, class R>
__device__ __forceinline__ int segment_class(const SweepParams<L, R>& p, int gy, int gz, int seg) {
  if (!p.seg_class) return 1;
  const uint32_t idx = (uint32_t)(gy + p.g.arr_ny * gz) * (uint32_t)p.nseg + (uint32_t)seg;
  const uint32_t word = ((const __attribute__((address_space(4))) uint32_t*)p.seg_class)[idx >> 2];
  return (int)((word >> (8u * (idx & 3u))) & 0xffu);
}

// Offsets (in elements) to the +-1 neighbours along each axis, with the
// optional in-kernel periodic wrap.  Real nodes are 1 .. lat-2.
struct AxisOff {
  int p, m;
};
__device__ __forceinline__ AxisOff axis_off(int c, int lat, int stride, int wrap) {
  AxisOff o;
  o.p = stride;
  o.m = -stride;
  if (wrap) {
    if (c == lat - 2) o.p = -(lat - 3) * stride;
    if (c == 1) o.m = (lat - 3) * stride;
  }
  return o;
}

template <class L, int I>
__device__ __forceinline__ int dir_offset(const AxisOff& ox, const AxisOff& oy, const AxisOff& oz, bool forward) {
  // forward: offset of x + e_i ; !forward: offset of x - e_i
  int off = 0;
  constexpr int ex = L::ex(I), ey = L::ey(I), ez = L::ez(I);
  if constexpr (ex != 0) off += ((ex > 0) == forward) ? ox.p : ox.m;
  if constexpr (ey != 0) off += ((ey > 0) == forward) ? oy.p : oy.m;
  if constexpr (ez != 0) off += ((ez > 0) == forward) ? oz.p : oz.m;
  return off;
}

// NT: bit 0 = non-temporal loads, bit 1 = non-temporal stores (populations are streamed exactly once per step)
template <int NT, class T>
__device__ __forceinline__ T ld(const T* p) {
  if constexpr (NT & 1) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int NT, class T>
__device__ __forceinline__ void st(T* p, T v) {
  if constexpr (NT & 2) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// Wave-uniform base address pinned to an SGPR pair.  The row kernels address every population as
// (uniform row base of direction i) + (the thread's x): with the base in SGPRs the access is
// `global_load/store v, v_x, s[base:base+1]` -- ONE shared 32-bit VGPR offset for all 19 + 19 streams instead of
// a 64-bit VGPR address per direction (which is what the compiler generates on its own: it re-associates the
// uniform and the per-lane parts).  That is ~40 VGPRs, i.e. the difference between 4 and 6-8 resident waves per
// SIMD, and resident waves are what the sweep's HBM rate tracks (profiles/r02/occupancy.md).  The empty asm
// keeps the compiler from folding the base back into per-lane arithmetic; the explicit global address space
// keeps the access a global_ (not flat_) instruction.  p MUST be wave-uniform.
#define SLF_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ SLF_GLOBAL T* uniform_base(T* p) {
  const uint64_t a = (uint64_t)p;
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);          // no-ops when the value already
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));  // lives in SGPRs
  asm("" : "+s"(lo), "+s"(hi));
  return (SLF_GLOBAL T*)(((uint64_t)hi << 32) | lo);
}
// A wave-uniform 32-bit value pinned to an SGPR (same trick; keeps the scalar address arithmetic scalar).
template <class T>
__device__ __forceinline__ T sgpr(T v) {
  static_assert(sizeof(T) == 4, "32-bit values only");
  uint32_t u = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, v));
  asm("" : "+s"(u));
  return __builtin_bit_cast(T, u);
}
// base + a per-lane BYTE offset kept in 32 bits (the saddr form takes a 32-bit VGPR offset).
template <class T>
__device__ __forceinline__ SLF_GLOBAL T* at_byte(SLF_GLOBAL T* base, uint32_t byte_off) {
  return (SLF_GLOBAL T*)((SLF_GLOBAL char*)base + byte_off);
}
template <class T>
__device__ __forceinline__ const SLF_GLOBAL T* at_byte(const SLF_GLOBAL T* base, uint32_t byte_off) {
  return (const SLF_GLOBAL T*)((const SLF_GLOBAL char*)base + byte_off);
}
template <int NT, class T>
__device__ __forceinline__ T ldg(const SLF_GLOBAL T* p) {
  if constexpr (NT & 1) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int NT, class T>
__device__ __forceinline__ void stg(SLF_GLOBAL T* p, T v) {
  if constexpr (NT & 2) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ---- x-face buffers ------------------------------------------------------------------------------------------
// Dense face buffers [z][k][y] over the padded (arr_ny x arr_nz) plane, k = position of the direction among the NXD
// with the same sign of e_x: a range of z-planes is one contiguous piece, which is what lets the halo stream send the
// planes a z-chunk of the sweep has completed while the next chunk computes (sailfish_amd/xface.py).
// Row of (y, z) and its +-1 neighbours along y / z with the same in-sweep wrap as the distribution arrays (values
// that leave through a face land in the row the neighbour's edge node sits in).
template <class L>
constexpr int count_x_dirs() {
  int n = 0;
  for (int i = 0; i < L::Q; i++) n += (L::ex(i) > 0) ? 1 : 0;
  return n;
}
struct FaceRows {
  int row;          // gy + NXD * arr_ny * gz
  AxisOff oy, oz;   // strides 1 and NXD * arr_ny
  int kstride;      // arr_ny
};
template <class L>
__device__ __forceinline__ FaceRows face_rows(const Geometry& g, int gy, int gz) {
  constexpr int NXD = count_x_dirs<L>();
  FaceRows fr;
  fr.row = gy + NXD * g.arr_ny * gz;
  fr.oy = axis_off(gy, g.lat_ny, 1, g.wrap[1]);
  fr.oz = (g.dim == 3) ? axis_off(gz, g.lat_nz, NXD * g.arr_ny, g.wrap[2]) : AxisOff{0, 0};
  fr.kstride = g.arr_ny;
  return fr;
}
// position of direction I among the directions with the same sign of e_x (ascending I)
template <class L, int I>
constexpr int x_dir_rank() {
  int n = 0;
  for (int i = 1; i < I; i++) n += ((L::ex(i) > 0) == (L::ex(I) > 0) && L::ex(i) != 0) ? 1 : 0;
  return n;
}
// element of direction I in the row of (y, z) [own] / of (y, z) + e_I (forward) / - e_I
template <class L, int I>
__device__ __forceinline__ int face_elem(const FaceRows& fr, int shift) {     // shift: 0 own row, +1 forward, -1 backward
  const AxisOff ox0 = {0, 0};
  const int off = (shift == 0) ? 0 : dir_offset<L, I>(ox0, fr.oy, fr.oz, shift > 0);
  return fr.row + off + fr.kstride * x_dir_rank<L, I>();
}
// "Nothing crossed the face here" = the all-ones bit pattern the buffers are cleared with (memset 0xFF), compared
// bit for bit: a NaN or an infinity that a diverging neighbour really sent is delivered like any other value, so the
// blow-up crosses the face and the invalid-value check trips on the receiving side too.
__device__ __forceinline__ bool face_value_present(float v) { return __builtin_bit_cast(uint32_t, v) != 0xffffffffu; }
__device__ __forceinline__ bool face_value_present(double v) { return __builtin_bit_cast(uint64_t, v) != ~0ull; }
// Incoming populations of an edge node: f_I with e_x > 0 at x = 1 come from the low neighbour, e_x < 0 at x = nx from
// the high one.  PULL = the odd AA step (the value sits in the row the pull reads from), otherwise the node's own row.
// Entries never written (the sender's edge node is excluded, or the entry belongs to a ghost row the neighbour does not
// sweep) leave f as loaded from the arrays.
template <class L, class R, bool PULL>
__device__ __forceinline__ void x_face_receive(const SweepParams<L, R>& p, R (&f)[L::Q], int x, int nx, const FaceRows& fr) {
  if (p.xrecv[0] && x == 1) {
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) > 0) {
        const R val = p.xrecv[0][face_elem<L, I>(fr, PULL ? -1 : 0)];
        if (face_value_present(val)) f[I] = val;
      }
    });
  }
  if (p.xrecv[1] && x == nx) {
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) < 0) {
        const R val = p.xrecv[1][face_elem<L, I>(fr, PULL ? -1 : 0)];
        if (face_value_present(val)) f[I] = val;
      }
    });
  }
}
// Outgoing populations of the even AA step: the post-collision f_I the neighbour's next (odd) step will pull.
template <class L, class R>
__device__ __forceinline__ void x_face_send_own_row(const SweepParams<L, R>& p, const R (&f)[L::Q], int x, int nx,
                                                    const FaceRows& fr) {
  if (p.xsend[1] && x == nx) {
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) > 0) p.xsend[1][face_elem<L, I>(fr, 0)] = f[I];
    });
  }
  if (p.xsend[0] && x == 1) {
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) < 0) p.xsend[0][face_elem<L, I>(fr, 0)] = f[I];
    });
  }
}

// Everything between loading the populations of a node and streaming them: macroscopic quantities,
// pre-collision boundary conditions, collision, half-way bounce-back stores (reference
// lb_single_fluid.mako:175-228).  Shared by all sweep kernels so that they differ only in access shape.
constexpr uint32_t INVALID_NODE = 0xffffffffu;
// the sweep of an indirectly addressed module over its slots (slf_slots.hip; called from slf_kernels.hip)
template <class L, class R>
hipError_t launch_slot_sweep(int model, int prop, int bc_level, const SweepParams<L, R>& q, hipStream_t s);
constexpr uint32_t OPTION_CHECK_INVALID = 4u;   // kernel `options`: 1 save macro fields, 2 bulk (reference), 4 this

// On-GPU invalid value check (reference checkInvalidValues, geo_helpers.mako:193-213, enabled by
// --check_invalid_results_gpu): the reference prints and traps; here a wet node whose density is not
// finite (any non-finite population makes it so) raises a flag and records the first position, and the
// host polls it with slf_module_poll_invalid() -- the run ends with backend.FatalError instead of a dead GPU.
template <class R>
__device__ __forceinline__ void check_invalid(uint32_t* status, uint32_t options, R rho, int x, int y, int z) {
  if ((options & OPTION_CHECK_INVALID) && !__builtin_isfinite(rho)) {
    if (atomicOr(status, 1u) == 0u) {
      status[1] = (uint32_t)x;
      status[2] = (uint32_t)y;
      status[3] = (uint32_t)z;
    }
  }
}

// gi: dense node index; si: the node's slot in the distribution arrays (= gi unless INDIRECT).
// BCL: the module's Geometry::bc_level the instantiation is for (2 = everything; the outflow, do-nothing and full-slip nodes
// exist at level 2 only, so that the level-1 kernels of the usual wall / inlet / outlet conditions do not carry their code).
// ROUNDOFF: the --minimize_roundoff formulation (slf_node.h: macro_roundoff, bgk_relax_roundoff): BGK; fluid, full-way
// and half-way bounce-back nodes (the module is refused otherwise); the per-node kernels only.
// TURB: the BGK collision with the options of the reference's relaxation preamble (--regularized, --subgrid; which of
// the two: SweepParams::turb_flags) -- slf_node.h bgk_relax_turb; the per-node kernels only, MODEL = BGK.
template <class L, class R, int MODEL, int PROP, bool GENERAL, bool INDIRECT = false, int FORCE = FORCE_RUNTIME, int BCL = 2,
          bool ROUNDOFF = false, bool TURB = false>
__device__ __forceinline__ void node_update(const SweepParams<L, R>& p, R (&f)[L::Q], uint32_t code, int kind,
                                            uint32_t gi, const AxisOff& ox, const AxisOff& oy, const AxisOff& oz,
                                            R& rho, R (&v)[3], bool& wet, uint32_t si = INVALID_NODE) {
  if constexpr (!INDIRECT) si = gi;
  const Geometry& g = p.g;
  const size_t ds = g.dist_size;
  (void)ds;
  if constexpr (ROUNDOFF) {
    wet = GENERAL ? kind_is_wet(kind) : true;
    // equilibrium density / velocity nodes (boundary.mako:425-459, 492-504 with config.minimize_roundoff: the sum of the
    // shifted populations is rho - 1; the imposed density enters as par_rho - 1; sym.py:573-682): the two boundary
    // conditions whose reference expressions are consistent under the option (the regularized / Zou-He nodes are not:
    // ex_eq_flux keeps multiplying by the density DELTA, DESIGN.md §9 -- refused at module creation)
    bool bc = false;
    if constexpr (GENERAL) {
      const int orientation = (int)(code >> g.orient_shift);
      if ((kind == NK_EQUILIBRIUM_DENSITY || kind == NK_EQUILIBRIUM_VELOCITY) && orientation != 0) {
        const int pidx = (int)((code >> g.param_shift) & g.param_mask);
        bc = true;
        with_orientation<L>(orientation, [&](auto O) {
          fill_missing_with_opposite<L, R, O>(f);
          const R rs = density<L, R>(f);
          if (kind == NK_EQUILIBRIUM_DENSITY) {
            const R par_rho = p.node_params[pidx];
            const R t = ((rs + (R)1) - par_rho) / par_rho;
            constexpr int n = L::dir2vecidx(O);
            v[0] = v[1] = v[2] = (R)0;
            static_for<0, L::dim>([&](auto D) {
              constexpr int e = e_comp<L>(n, D);
              if constexpr (e > 0) v[D] = (R)0 - t;
              if constexpr (e < 0) v[D] = t;
            });
            rho = par_rho - (R)1;
          } else {
            v[0] = p.node_params[pidx];
            v[1] = p.node_params[pidx + 1];
            v[2] = (R)0;
            if constexpr (L::dim == 3) v[2] = p.node_params[pidx + 2];
            const R nv = ndotv<L, R, O>(v);
            rho = (rs + nv) / ((R)1 - nv);
          }
        });
      }
    }
    if (!bc) macro_roundoff<L, R>(f, rho, v);
    if (GENERAL && (kind == NK_EQUILIBRIUM_DENSITY || kind == NK_EQUILIBRIUM_VELOCITY)) {
      set_equilibrium<L, R>(f, rho, rho + (R)1, v);          // boundary.mako:797-809 with rho0 = rho + 1
    }
    if (GENERAL && kind == NK_FULL_BB) bounce_back<L, R>(f);
    if (wet && p.relaxation_enabled)
      bgk_relax_roundoff<L, R>(f, rho, v, p.cp.omega, p.cp.guo_pref, p.cp.has_force != 0, p.cp.accel, p.cp.force_edm != 0);
    if (GENERAL && kind == NK_HALF_BB) {
      const int orientation = (int)(code >> g.orient_shift);
      static_for<1, L::Q>([&](auto I) {
        bool missing;
        if (g.use_link_tags) {
          missing = ((orientation >> (I - 1)) & 1) == 0;
        } else {
          missing = false;
          with_orientation<L>(orientation, [&](auto O) {
            if constexpr (is_missing<L, L::opp(I), O>()) missing = true;
          });
        }
        if (missing) {
          if constexpr (PROP == PROP_AA_EVEN) {
            const int off = dir_offset<L, I>(ox, oy, oz, true);
            uint32_t t = (uint32_t)((int)gi + off);
            if constexpr (INDIRECT) t = p.nodes[t];
            if (!INDIRECT || t != INVALID_NODE) (p.dout + ds * (size_t)I)[t] = f[I];
          } else {
            (p.dout + ds * (size_t)L::opp(I))[si] = f[I];
          }
        }
      });
    }
    return;
  }
  if constexpr (GENERAL && BCL == 0) {
    wet = kind_is_wet(kind);
    macro_standard<L, R>(f, p.cp.incompressible != 0, rho, v);
    if (kind == NK_FULL_BB) bounce_back<L, R>(f);
    if (wet && p.relaxation_enabled) {
      if constexpr (MODEL == 0) bgk_relax<L, R, FORCE>(f, rho, v, p.cp);
      else mrt_relax<L, R, FORCE>(f, v, p.cp, false);
    }
  } else if constexpr (GENERAL) {
    wet = kind_is_wet(kind);
    const int orientation = (int)(code >> g.orient_shift);
    const int pidx = (int)((code >> g.param_shift) & g.param_mask);
    const bool inc = p.cp.incompressible != 0;
    // ---- fixMissingDistributions (boundary.mako:509-603): outflow nodes fill their unknown populations from
    // the incoming state of the nodes one / two steps along the inward normal (two-copy pattern only)
    if constexpr (PROP == PROP_AB && BCL == 2) {
      if ((kind == NK_COPY || kind == NK_YU_OUTFLOW) && orientation != 0) {
        with_orientation<L>(orientation, [&](auto O) {
          constexpr int n = L::dir2vecidx(O);
          const int off1 = dir_offset<L, n>(ox, oy, oz, true);
          const int off2 = 2 * (L::ex(n) + L::ey(n) * g.arr_nx + L::ez(n) * g.arr_nxy);
          uint32_t s1 = (uint32_t)((int)gi + off1), s2 = (uint32_t)((int)gi + off2);
          if constexpr (INDIRECT) {
            s1 = p.nodes[s1];
            s2 = p.nodes[s2];
          }
          static_for<1, L::Q>([&](auto I) {
            if constexpr (is_missing<L, I, O>()) {
              const R* d = p.din + ds * (size_t)I;
              if (kind == NK_COPY) {
                if (!INDIRECT || s1 != INVALID_NODE) f[I] = d[s1];
              } else {
                if (!INDIRECT || (s1 != INVALID_NODE && s2 != INVALID_NODE)) f[I] = (R)2 * d[s1] - d[s2];
              }
            }
          });
        });
      }
    }
    // ---- macroscopic quantities (getMacro, boundary.mako:465-507)
    const bool density_bc = kind == NK_EQUILIBRIUM_DENSITY || kind == NK_ZOUHE_DENSITY || kind == NK_REGULARIZED_DENSITY;
    const bool bc_macro = (kind == NK_REGULARIZED_VELOCITY || kind == NK_EQUILIBRIUM_VELOCITY ||
                           kind == NK_ZOUHE_VELOCITY || density_bc) && orientation != 0;
    if (!bc_macro) {
      macro_standard<L, R>(f, inc, rho, v);
    } else if (density_bc) {
      with_orientation<L>(orientation, [&](auto O) { macro_density_bc<L, R, O>(f, p.node_params[pidx], rho, v); });
      if (kind == NK_ZOUHE_DENSITY) {
        // boundary.mako:487-494: fix the populations, take u from them, keep the imposed density
        const R par_rho = rho;
        with_orientation<L>(orientation, [&](auto O) { zouhe_bb<L, R, O>(f, par_rho, inc ? (R)1 : par_rho, v); });
        macro_standard<L, R>(f, inc, rho, v);
        rho = par_rho;
      }
    } else {
      with_orientation<L>(orientation,
                          [&](auto O) { macro_velocity_bc<L, R, O>(f, p.node_params + pidx, inc, rho, v); });
    }
    // ---- pre-collision boundary conditions (boundary.mako:784-878)
    const R rho0 = inc ? (R)1 : rho;
    if (kind == NK_FULL_BB) {
      bounce_back<L, R>(f);
    } else if (kind == NK_EQUILIBRIUM_DENSITY || kind == NK_EQUILIBRIUM_VELOCITY) {
      set_equilibrium<L, R>(f, rho, rho0, v);
    } else if (kind == NK_REGULARIZED_VELOCITY || kind == NK_REGULARIZED_DENSITY) {
      if (orientation == 0) {
        bounce_back<L, R>(f);  // nt_dir_other fallback, boundary.mako:336-338
      } else {
        with_orientation<L>(orientation, [&](auto O) { regularized_bc<L, R, O>(f, rho, rho0, v); });
      }
    } else if (kind == NK_ZOUHE_VELOCITY) {
      if (orientation == 0) {
        bounce_back<L, R>(f);
      } else {
        with_orientation<L>(orientation, [&](auto O) { zouhe_bb<L, R, O>(f, rho, rho0, v); });
      }
    }
    if constexpr (BCL == 2) {
      if (kind == NK_SLIP) slip_reflect<L, R>(f, orientation);   // boundary.mako:837-855
    }
    // NTDoNothing, in-place pattern (boundary.mako:862-876): the unknown populations keep their value -- the node stores
    // what it has just read to where its NEXT step reads it: an odd step (the next one reads the node's own slots) into its
    // own slot, an even step (the next one pulls from the neighbours' opposite slots) into the opposite slot of the node
    // the population would have come from -- the ghost node behind the boundary.  Nobody else touches either slot.
    if constexpr (PROP != PROP_AB && !INDIRECT && BCL == 2) {
      if (kind == NK_DO_NOTHING) {
        with_orientation<L>(orientation, [&](auto O) {
          static_for<1, L::Q>([&](auto I) {
            if constexpr (is_missing<L, I, O>()) {
              if constexpr (PROP == PROP_AA_ODD) {
                (p.dout + ds * (size_t)I)[si] = f[I];
              } else {
                constexpr int J = L::opp(I);
                const int off = dir_offset<L, J>(ox, oy, oz, true);
                (p.dout + ds * (size_t)J)[(uint32_t)((int)gi + off)] = f[I];
              }
            }
          });
        });
      }
    }
    // ---- collision (relaxate, relaxation.mako:196-202: wet nodes only)
    if (wet && p.relaxation_enabled) {
      if constexpr (MODEL == 0 && TURB) {
        bgk_relax_turb<L, R>(f, rho, v, p.cp, p.turb_flags, p.turb_visc, p.turb_c2x36);
      } else if constexpr (MODEL == 0) {
        bgk_relax<L, R, FORCE>(f, rho, v, p.cp);
      } else {
        mrt_relax<L, R, FORCE>(f, v, p.cp, kind == NK_EQUILIBRIUM_DENSITY || kind == NK_EQUILIBRIUM_VELOCITY);
      }
    }
    // ---- post-collision: half-way bounce-back (boundary.mako:653-683)
    if (kind == NK_HALF_BB) {
      static_for<1, L::Q>([&](auto I) {
        bool missing;
        if (g.use_link_tags) {
          missing = ((orientation >> (I - 1)) & 1) == 0;  // direction I points to a non-fluid node
        } else {
          missing = false;
          with_orientation<L>(orientation, [&](auto O) {
            if constexpr (is_missing<L, L::opp(I), O>()) missing = true;
          });
        }
        if (missing) {
          // population opp(I) is undefined here: feed it with the reflected f_I.
          if constexpr (PROP == PROP_AA_EVEN) {
            const int off = dir_offset<L, I>(ox, oy, oz, true);
            uint32_t t = (uint32_t)((int)gi + off);
            if constexpr (INDIRECT) t = p.nodes[t];
            if (!INDIRECT || t != INVALID_NODE) (p.dout + ds * (size_t)I)[t] = f[I];
          } else {
            (p.dout + ds * (size_t)L::opp(I))[si] = f[I];
          }
        }
      });
    }
  } else {
    macro_standard<L, R>(f, p.cp.incompressible != 0, rho, v);
    if (p.relaxation_enabled) {
      if constexpr (MODEL == 0 && TURB) bgk_relax_turb<L, R>(f, rho, v, p.cp, p.turb_flags, p.turb_visc, p.turb_c2x36);
      else if constexpr (MODEL == 0) bgk_relax<L, R, FORCE>(f, rho, v, p.cp);
      else mrt_relax<L, R, FORCE>(f, v, p.cp, false);
    }
  }

}

template <class L, class R>
inline SweepParams<L, R> make_params(const Geometry& g, const Physics& ph, const SweepArgs& a, int y0, int z0) {
  SweepParams<L, R> p;
  p.nodes = (const uint32_t*)a.nodes;
  p.map = (const uint32_t*)a.map;
  p.din = (const R*)a.dist_in;
  p.dout = (R*)a.dist_out;
  p.rho = (R*)a.rho;
  p.vx = (R*)a.v[0];
  p.vy = (R*)a.v[1];
  p.vz = (R*)a.v[2];
  p.node_params = (const R*)a.node_params;
  p.status = (uint32_t*)a.status;
  p.options = a.options;
  for (int k = 0; k < 2; k++) {
    p.xsend[k] = (R*)a.xsend[k];
    p.xrecv[k] = (const R*)a.xrecv[k];
  }
  p.y0 = y0;
  p.z0 = z0;
  p.y1 = p.z1 = 0;
  p.seg_class = nullptr;
  p.row_class = nullptr;
  p.bc_rows = nullptr;
  p.nseg = 0;
  p.row_mode = 0;
  p.slot_gi = p.slot_yz = nullptr;
  p.n_slots = 0;
  if (a.slots && a.slots->nodes == a.nodes && a.nodes) {
    p.slot_gi = a.slots->slot_gi;
    p.slot_yz = a.slots->slot_yz;
    p.n_slots = a.slots->n_slots;
  }
  if (a.rows && a.rows->map == a.map && a.map) {
    p.seg_class = a.rows->seg_class;
    p.row_class = a.rows->row_class;
    p.bc_rows = a.rows->bc_rows;
    p.nseg = a.rows->nseg;
  }
  p.relaxation_enabled = ph.relaxation_enabled;
  p.turb_flags = (ph.regularized ? 1 : 0) | (ph.subgrid ? 2 : 0);
  p.turb_visc = (R)ph.visc;
  p.turb_c2x36 = (R)(36.0 * ph.smagorinsky_const * ph.smagorinsky_const);
  p.g = g;
  p.cp.omega = (R)(1.0 / ph.tau);
  for (int k = 0; k < L::Q; k++) p.cp.mrt_s[k] = (R)ph.mrt_rates[k];
  for (int d = 0; d < 3; d++) p.cp.accel[d] = (R)ph.accel[d];
  p.cp.guo_pref = (R)(3.0 * (1.0 - 0.5 / ph.tau));
  p.cp.incompressible = ph.incompressible;
  p.cp.has_force = ph.has_force;
  p.cp.force_edm = ph.force_edm;
  return p;
}


// Returns true when the launch was handled by a tuned kernel (status in *err).
bool launch_sweep_fast(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph, const SweepArgs& a,
                       int y0, int y1, int z0, int z1, int block_x, hipStream_t s, hipError_t* err);
// Whole-row kernels for every lattice / precision / model, with or without the node map (slf_row.hip).
bool launch_sweep_row(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph, const SweepArgs& a,
                      int y0, int y1, int z0, int z1, hipStream_t s, hipError_t* err);

}  // namespace slf
