// Binary-fluid Shan-Chen model (two lattices coupled by a pseudopotential force) for gfx950.
//
//   sc_macro_kernel     ShanChenPrepareMacroFields      (reference templates/models/binary_shan_chen.mako:19-87)
//   sc_sweep_kernel<K>  ShanChenCollideAndPropagate{0,1} (binary_shan_chen.mako:89-141, shan_chen.mako:9-84)
//   sc_fused_kernel     both of them in one pass ("ShanChenCollideAndPropagateFused"): rho, phi, u and the 18-point
//                       pseudopotential stencil of each field are read once for the two lattices;
//                       <OWNV> ("...FusedV"): densities and velocity of the node formed in the sweep, lattice 1 parked in LDS
//   sc_macro_kernel<VOUT = false>, sc_density_pull_kernel   "ShanChenPrepareDensities": rho and phi only (+ u on output steps)
//   sc_pull_rows        the odd in-place step's populations through aligned loads and a one-lane shift
//   sc_init_kernel      SetInitialConditions (binary)    (templates/models/lb_binary_fluid.mako:87-127)
//
// Same layout, streaming modes and launch shape as the single-fluid sweep (slf_kernels.hip).  The force
//   F_k(x) = - sum_j G_kj psi(rho_k(x)) sum_i w_i e_i psi(rho_j(x + e_i))
// reads the 18 neighbour densities of the *other* field through the cache hierarchy (each value is
// reused by 18 nodes; rows of a workgroup are contiguous), the acceleration F_k / rho_k enters the BGK
// collision through Guo forcing exactly like a body force (relaxation_common.mako:56-64,110-149).
#include "slf_rowpush.h"

namespace slf {

template <class L, class R>
struct ScParams {
  const uint32_t* __restrict__ nodes;   // indirect addressing: dense node -> slot of the distribution arrays, else NULL
  const uint32_t* __restrict__ map;
  const R* d_in;     // sweep: lattice in      | macro: lattice 0
  R* d_out;          // sweep: lattice out     | macro: lattice 1 (read)
  R* rho0;
  R* rho1;
  R* vx;
  R* vy;
  R* vz;
  uint32_t options;
  int y0, z0;
  Geometry g;
  R omega[2];        // 1/tau, 1/tau_phi
  R guo_pref[2];
  R G[2];            // couplings of this lattice with field 0 and field 1 (sweep only)
  R accel[3];        // additional body-force acceleration
  int has_body_force;
  // fused sweep of both lattices: the second lattice and its constants (d_in / d_out / G / accel = lattice 0)
  const R* d_in2;
  R* d_out2;
  R G2[2];
  R accel2[3];
  int has_body_force2;
  int potential;
  int force_edm;
  int xcd_shift;     // xcd_row(): rows per XCD and block of rows = 1 << this
  // connected x faces (XF instantiations; slf_module_set_xface_planes): dense planes instead of ghost columns
  R* xsend[2][2];          // [lattice][low / high face]: populations that leave, [z][k][y] as slf_sweep.h
  const R* xrecv[2][2];    //                             populations that enter
  R* msend[2];             // [face]: densities of my first / last column, [z][field][y]
  const R* mrecv[2];       //         densities of the neighbour's last / first column
};

template <class R, int POT>
__device__ __forceinline__ R sc_psi(R rho) {
  // sym.py:896-908: linear psi = rho; classic psi = 1 - exp(-rho)
  if constexpr (POT == 0) return rho;
  else if constexpr (sizeof(R) == 4) return 1.0f - expf(0.0f - rho);      // the reference's exp() on a float argument
  else return (R)1 - exp((R)0 - rho);
}

// The module's potential (a run-time constant of the launch) as a compile-time constant of `body`: ONE wave-uniform
// branch around a whole block of psi evaluations.  With the choice inside sc_psi() the compiler turned it into a select:
// every one of the 38 psi values of a node went through the 15-instruction expf sequence and was thrown away again when
// the potential is linear -- a third of the fused sweep's vector instructions (1577 per wave, the vector ALUs ~80 % busy:
// profiles/r05/sq_summary_shan_chen_before.txt).
template <class F>
__device__ __forceinline__ void sc_with_potential(int potential, F&& body) {
  if (potential == 0) body(std::integral_constant<int, 0>{});
  else body(std::integral_constant<int, 1>{});
}

// Cache hints for the populations: streamed once per kernel in 3-D (non-temporal); 2-D lattices live in the caches.
template <class L>
constexpr int sc_nt() { return L::dim == 3 ? 3 : 0; }

// A node's row and lane position, the way slf_row.hip addresses memory: everything that is the same for the whole
// workgroup (y, z, the row's first element, the y / z neighbour offsets) is pinned to SGPRs, so that every access is
// (uniform base) + (one shared 32-bit lane offset) -- see uniform_base(), slf_sweep.h.
struct ScNode {
  int gx, gy, gz;
  uint32_t row, xi, gi;
  AxisOff ox, oy, oz;
};
template <class L>
__device__ __forceinline__ ScNode sc_node(const Geometry& g, int y0, int z0, int nx, bool& live, int xcd_shift = 0) {
  ScNode n;
  // rows that share rho / phi rows through the L2 go to one XCD (xcd_row(), slf_sweep.h)
  const int by = xcd_row((int)blockIdx.y, xcd_shift);
  n.gy = sgpr(y0 + by);
  n.gz = (L::dim == 3) ? sgpr(z0 + (int)blockIdx.z) : 0;
  n.gx = 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  live = n.gx <= nx;
  n.row = sgpr((uint32_t)g.arr_nx * (uint32_t)n.gy + (uint32_t)g.arr_nxy * (uint32_t)n.gz);
  n.xi = (uint32_t)(live ? n.gx : nx);      // idle lanes: an in-row address, never stored
  n.gi = n.row + n.xi;
  n.ox = axis_off(n.gx, g.lat_nx, 1, g.wrap[0]);
  n.oy = axis_off(n.gy, g.lat_ny, g.arr_nx, g.wrap[1]);
  n.oz = (L::dim == 3) ? axis_off(n.gz, g.lat_nz, g.arr_nxy, g.wrap[2]) : AxisOff{0, 0};
  n.oy.p = sgpr(n.oy.p); n.oy.m = sgpr(n.oy.m); n.oz.p = sgpr(n.oz.p); n.oz.m = sgpr(n.oz.m);
  return n;
}

// element (row + yz-offset of direction I, forward or backward) + (xi shifted along x) of a node-indexed array
template <class L, int I, class T>
__device__ __forceinline__ const SLF_GLOBAL T* sc_neighbour(const T* base, const ScNode& n, bool forward) {
  const AxisOff ox0 = {0, 0};
  const int off = dir_offset<L, I>(ox0, n.oy, n.oz, forward);          // y, z part: uniform
  constexpr int ex = L::ex(I);
  const int xs = (ex == 0) ? 0 : (((ex > 0) == forward) ? n.ox.p : n.ox.m);   // x part: per lane
  return at_byte(uniform_base(base + (uint32_t)((int)n.row + off)), (uint32_t)((int)n.xi + xs) * (uint32_t)sizeof(T));
}

// INDIRECT (reference subdomain_runner.py:829-878, kernel_common.mako:140-167; serves NNSubdomainRunner too): the
// distribution arrays hold the active nodes only; the node's own slot and the slots of its neighbours come from the
// dense table `nodes`, the fields rho / phi / v and the node map stay dense.
template <class L, class R, int PROP, bool INDIRECT = false>
__device__ __forceinline__ void sc_load(R (&f)[L::Q], const R* din, size_t ds, const ScNode& n,
                                        const uint32_t* nodes = nullptr, uint32_t si = 0) {
  if constexpr (INDIRECT) {
    static_for<0, L::Q>([&](auto I) {
      if constexpr (PROP == PROP_AA_ODD) {
        const uint32_t sn = nodes[(uint32_t)((int)n.gi + dir_offset<L, I>(n.ox, n.oy, n.oz, false))];
        f[I] = (sn != INVALID_NODE) ? (din + ds * (size_t)L::opp(I))[sn] : (R)0;
      } else {
        f[I] = (din + ds * (size_t)I)[si];
      }
    });
    return;
  }
  static_for<0, L::Q>([&](auto I) {
    if constexpr (PROP == PROP_AA_ODD) {
      constexpr int nt = (L::ex(I) != 0) ? (sc_nt<L>() & ~1) : sc_nt<L>();     // shifted pulls share their lines
      f[I] = ldg<nt>(sc_neighbour<L, I>(din + ds * (size_t)L::opp(I), n, false));
    } else {
      f[I] = ldg<sc_nt<L>()>(at_byte(uniform_base(din + ds * (size_t)I + n.row), n.xi * (uint32_t)sizeof(R)));
    }
  });
}

// Aligned pull of the populations of BOTH lattices for one (y, z) row in the odd in-place step -- the mirror image of
// row_push(): node x needs slot opp(i) of node x - e_i; the lane that owns x' loads that slot of (y, z) - e_i(yz) at
// ITS OWN x' (whole 128-byte lines, every line fetched by exactly one wave) and the value travels one lane along x: a
// DPP wave shift, one LDS word per direction and lattice between neighbouring waves, the periodic wrap as the cyclic
// continuation.  For rows that are wrapped along x inside the kernel and owned by one workgroup (nx <= 1024); every
// thread of the workgroup must call it (one barrier).  Pure data movement: the same values as sc_load<PROP_AA_ODD>.
template <class L, class R>
__device__ __forceinline__ void sc_pull_rows(R (&fa)[L::Q], R (&fb)[L::Q], const R* da, const R* db, size_t ds,
                                             const ScNode& n, int nx) {
  constexpr int NW = 16;
  constexpr int NXD = count_x_dirs<L>();
  __shared__ R s_p[2][NW + 1][NXD], s_m[2][NW + 1][NXD];     // [lattice][wave | NW = the wrap][direction]
  const int lane = (int)threadIdx.x & 63;
  const int w = sgpr((int)threadIdx.x >> 6);
  const bool is_xe = n.gx == nx;
  const AxisOff ox0 = {0, 0};
  static_for<0, L::Q>([&](auto I) {
    const int off = dir_offset<L, I>(ox0, n.oy, n.oz, false);        // y, z part of x - e_i: the same for the whole row
    const uint32_t lo = n.xi * (uint32_t)sizeof(R);
    fa[I] = ldg<sc_nt<L>()>(at_byte(uniform_base(da + ds * (size_t)L::opp(I) + (uint32_t)((int)n.row + off)), lo));
    fb[I] = ldg<sc_nt<L>()>(at_byte(uniform_base(db + ds * (size_t)L::opp(I) + (uint32_t)((int)n.row + off)), lo));
  });
  // what leaves a wave through its last / first lane
  if (lane == 63 || is_xe) {
    const int slot = is_xe ? NW : w;
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) > 0) {
        constexpr int k = x_dir_rank<L, I>();
        s_p[0][slot][k] = fa[I];
        s_p[1][slot][k] = fb[I];
        if (is_xe && lane == 63) { s_p[0][w][k] = fa[I]; s_p[1][w][k] = fb[I]; }
      }
    });
  }
  if (lane == 0) {
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) < 0) {
        constexpr int k = x_dir_rank<L, I>();
        s_m[0][w][k] = fa[I];
        s_m[1][w][k] = fb[I];
      }
    });
  }
  __syncthreads();
  const int from_p = (w == 0) ? NW : w - 1;
  static_for<1, L::Q>([&](auto I) {
    constexpr int k = x_dir_rank<L, I>();
    if constexpr (L::ex(I) > 0) {                      // from x - 1
      fa[I] = lane_shift1<R, true>(s_p[0][from_p][k], fa[I]);
      fb[I] = lane_shift1<R, true>(s_p[1][from_p][k], fb[I]);
    } else if constexpr (L::ex(I) < 0) {               // from x + 1; x = nx (need not be lane 63) takes x = 1
      const R ta = lane_shift1<R, false>(s_m[0][w + 1][k], fa[I]);
      const R tb = lane_shift1<R, false>(s_m[1][w + 1][k], fb[I]);
      fa[I] = is_xe ? s_m[0][0][k] : ta;
      fb[I] = is_xe ? s_m[1][0][k] : tb;
    }
  });
}

// Shan-Chen acceleration of one lattice (sc_calculate_force, shan_chen.mako:9-27): the 18 neighbour values of each
// coupled field come through the caches (every value is used by 18 nodes).
template <class L, class R, int NFIELDS>
__device__ __forceinline__ void sc_accel(const R* const (&fields)[2], const R (&G)[2], R rho, int potential, const ScNode& n,
                                         R (&a)[3]) {
  sc_with_potential(potential, [&](auto POT) {
    static_for<0, NFIELDS>([&](auto J) {
      const R cc = G[J];
      if (cc != (R)0) {
        R force[3] = {(R)0, (R)0, (R)0};
        static_for<1, L::Q>([&](auto I) {
          const R psi = sc_psi<R, POT>(*sc_neighbour<L, I>(fields[J], n, true));
          static_for<0, L::dim>([&](auto D) {
            constexpr int e = e_comp<L>(I, D);
            if constexpr (e > 0) force[D] = force[D] + psi * Weights<L, R>::w(I);
            if constexpr (e < 0) force[D] = force[D] + psi * ((R)0 - Weights<L, R>::w(I));
          });
        });
        const R psi_loc = sc_psi<R, POT>(rho);
        static_for<0, L::dim>([&](auto D) {
          force[D] = force[D] * (((R)0 - psi_loc) * cc);
          a[D] = a[D] + force[D];
        });
      }
    });
  });
}

// the streaming part of a sweep: whole-row push (3-D x-streaming steps), own slots (even AA step), or per-node push
template <class L, class R, int PROP, bool GENERAL, bool ROW, bool INDIRECT = false>
__device__ __forceinline__ void sc_store(const Geometry& g, R (&f)[L::Q], R* dout, size_t ds, const ScNode& n, int nx,
                                         bool live, bool active, const uint32_t* nodes = nullptr, uint32_t si = 0) {
  if constexpr (INDIRECT) {
    static_for<0, L::Q>([&](auto I) {
      if constexpr (PROP == PROP_AA_EVEN) {
        (dout + ds * (size_t)L::opp(I))[si] = f[I];
      } else {
        const uint32_t t = nodes[(uint32_t)((int)n.gi + dir_offset<L, I>(n.ox, n.oy, n.oz, true))];
        if (t != INVALID_NODE) (dout + ds * (size_t)I)[t] = f[I];
      }
    });
  } else if constexpr (ROW && PROP != PROP_AA_EVEN) {
    row_push<L, R, GENERAL, sc_nt<L>()>(g, f, dout, ds, n.row, n.xi, n.gx, nx, live, active, n.oy, n.oz);
  } else {
    static_for<0, L::Q>([&](auto I) {
      if constexpr (PROP == PROP_AA_EVEN) {
        stg<sc_nt<L>()>(at_byte(uniform_base(dout + ds * (size_t)L::opp(I) + n.row), n.xi * (uint32_t)sizeof(R)), f[I]);
      } else {
        const int off = dir_offset<L, I>(n.ox, n.oy, n.oz, true);
        (dout + ds * (size_t)I)[(uint32_t)((int)n.gi + off)] = f[I];
      }
    });
  }
}

// ---- connected x faces (XF): the binary model over the x-face planes of a 1-D decomposition along x ----
// Populations: per lattice the single-fluid scheme (slf_sweep.h x_face_receive / x_face_send_own_row, row_push's xsend).
// Densities: the force stencil of an edge node reads the neighbour subdomain's first / last column; the pass in front
// (sc_macro_kernel) stores rho and phi of ITS edge nodes into a send plane [z][field][y] over the padded (arr_ny x arr_nz)
// plane, and the sweep's edge lanes take the five values per field that sit across the face from the receive plane.  The
// density planes carry no "nothing crossed here" marker: an entry is either rewritten every step (a node whose density the
// pass in front forms) or never (a node that pass skips, a ghost row), and the caller fills every entry once from the
// fields as they are -- before the first step and after every host-side write of the state (xface.NNPlanes.prime): what a
// ghost column would have held.
template <class L, class R, bool PULL>
__device__ __forceinline__ void sc_face_receive(const R* const (&xrecv)[2], R (&f)[L::Q], int x, int nx, const FaceRows& fr) {
  // PULL (the odd in-place step): the value sits in the row the pull reads from, (y, z) - e_I; else in the node's own row
  if (xrecv[0] && x == 1) {
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) > 0) {
        const R val = xrecv[0][face_elem<L, I>(fr, PULL ? -1 : 0)];
        if (face_value_present(val)) f[I] = val;
      }
    });
  }
  if (xrecv[1] && x == nx) {
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) < 0) {
        const R val = xrecv[1][face_elem<L, I>(fr, PULL ? -1 : 0)];
        if (face_value_present(val)) f[I] = val;
      }
    });
  }
}
// the even in-place step: the post-collision values the neighbour's next (odd) step pulls across the face, own row
template <class L, class R>
__device__ __forceinline__ void sc_face_send_own_row(R* const (&xsend)[2], const R (&f)[L::Q], int x, int nx, const FaceRows& fr) {
  if (xsend[1] && x == nx) {
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) > 0) xsend[1][face_elem<L, I>(fr, 0)] = f[I];
    });
  }
  if (xsend[0] && x == 1) {
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) < 0) xsend[0][face_elem<L, I>(fr, 0)] = f[I];
    });
  }
}
// Odd in-place step next to a connected x face, fluid-only instantiations: what the edge node would pull out of the
// ghost column arrives through the plane (sc_face_receive overrides it), so its lane pulls from its own x -- a line the
// wave fetches anyway -- instead of from the ghost column, a line of its own per direction and row: +29 % reads on the
// 64-node rows of a four-way split of 256^3 (profiles/r06/pmc_summary_scs_4x_planes_before_skip.txt).  As slf_row.hip's
// SKIP_GHOST_PULL, and under the same condition: rows whose y / z neighbours are real or wrapped rows (every entry they
// read is then written by the neighbour's even step -- or was filled by NNPlanes.prime_own after a host-side write).
__device__ __forceinline__ ScNode sc_pull_node(const Geometry& g, const ScNode& n, int nx, bool recv_lo, bool recv_hi) {
  ScNode m = n;
  const bool inner = (g.wrap[1] || (n.gy > 1 && n.gy < g.lat_ny - 2)) && (g.dim < 3 || g.wrap[2] || (n.gz > 1 && n.gz < g.lat_nz - 2));
  if (inner) {
    if (recv_lo && n.gx == 1) m.ox.m = 0;
    if (recv_hi && n.gx == nx) m.ox.p = 0;
  }
  return m;
}

// row of (y, z) in a density plane, and the offsets to its y / z neighbours (wrapped like the arrays)
struct ScMacroRows {
  int row;
  AxisOff oy, oz;
  int fstride;      // field 1 sits this far behind field 0
};
__device__ __forceinline__ ScMacroRows sc_macro_rows(const Geometry& g, int gy, int gz) {
  ScMacroRows mr;
  mr.row = gy + 2 * g.arr_ny * gz;
  mr.oy = axis_off(gy, g.lat_ny, 1, g.wrap[1]);
  mr.oz = axis_off(gz, g.lat_nz, 2 * g.arr_ny, g.wrap[2]);
  mr.fstride = g.arr_ny;
  return mr;
}
// offset (in a density plane) of the k-th direction with e_x > 0 (PLUS) resp. < 0 from the row of (y, z): wave-uniform
template <class L, bool PLUS>
__device__ __forceinline__ void sc_macro_offsets(const ScMacroRows& mr, int (&off)[count_x_dirs<L>()]) {
  const AxisOff ox0 = {0, 0};
  static_for<1, L::Q>([&](auto I) {
    if constexpr (L::ex(I) != 0 && (L::ex(I) > 0) == PLUS) off[x_dir_rank<L, I>()] = dir_offset<L, I>(ox0, mr.oy, mr.oz, true);
  });
}

// What an edge lane knows about its face: which face, its receive plane, and where the five stencil values per field lie.
template <class L, class R>
struct ScEdge {
  bool lo, hi;
  const R* plane;
  int off[count_x_dirs<L>()];
  int fstride;
};
template <class L, class R>
__device__ __forceinline__ ScEdge<L, R> sc_edge(const ScParams<L, R>& p, const ScNode& n, int nx, bool live, ScNode& ns) {
  const Geometry& g = p.g;
  ScEdge<L, R> e;
  e.lo = live && p.mrecv[0] && n.gx == 1;
  e.hi = live && p.mrecv[1] && n.gx == nx;
  ns = n;
  if (e.lo) ns.ox.m = 0;          // the lane's own loads stay inside the row
  if (e.hi) ns.ox.p = 0;
  const ScMacroRows mr = sc_macro_rows(g, n.gy, n.gz);
  int offp[count_x_dirs<L>()], offm[count_x_dirs<L>()];
  sc_macro_offsets<L, true>(mr, offp);
  sc_macro_offsets<L, false>(mr, offm);
  static_for<0, count_x_dirs<L>()>([&](auto K) { e.off[K] = mr.row + (e.lo ? offm[K] : offp[K]); });
  e.plane = e.lo ? p.mrecv[0] : p.mrecv[1];
  e.fstride = mr.fstride;
  return e;
}
// sc_accel() of one field for a node that may sit on a connected x face (single-component model): the same sum in the
// same order, the values across the face from the receive plane
template <class L, class R>
__device__ __forceinline__ void sc_accel_edge(const R* field, R G, R rho, int potential, const ScNode& ns,
                                              const ScEdge<L, R>& e, R (&a)[3]) {
  sc_with_potential(potential, [&](auto POT) {
    if (G != (R)0) {
      R ev[count_x_dirs<L>()];
      static_for<0, count_x_dirs<L>()>([&](auto K) { ev[K] = (R)0; });
      if (e.lo || e.hi) {
        static_for<0, count_x_dirs<L>()>([&](auto K) { ev[K] = e.plane[e.off[K]]; });
      }
      R force[3] = {(R)0, (R)0, (R)0};
      static_for<1, L::Q>([&](auto I) {
        R nb = *sc_neighbour<L, I>(field, ns, true);
        if constexpr (L::ex(I) > 0) nb = e.hi ? ev[x_dir_rank<L, I>()] : nb;
        if constexpr (L::ex(I) < 0) nb = e.lo ? ev[x_dir_rank<L, I>()] : nb;
        const R psi = sc_psi<R, POT>(nb);
        static_for<0, L::dim>([&](auto D) {
          constexpr int ec = e_comp<L>(I, D);
          if constexpr (ec > 0) force[D] = force[D] + psi * Weights<L, R>::w(I);
          if constexpr (ec < 0) force[D] = force[D] + psi * ((R)0 - Weights<L, R>::w(I));
        });
      });
      const R psi_loc = sc_psi<R, POT>(rho);
      static_for<0, L::dim>([&](auto D) {
        force[D] = force[D] * (((R)0 - psi_loc) * G);
        a[D] = a[D] + force[D];
      });
    }
  });
}

// VOUT = false ("ShanChenPrepareDensities", the partner of the fused sweep that forms the velocity itself): only the
// two densities are stored -- three of the five written streams gone; the velocity arrays are brought up to date by
// the launches whose options ask for output (bit 0), which run the VOUT = true instantiation.
template <class L, class R, int PROP, bool GENERAL, bool INDIRECT = false, bool VOUT = true, bool XF = false>
__global__ void __launch_bounds__(1024) sc_macro_kernel(const ScParams<L, R> p) {
  static_assert(!XF || !INDIRECT, "x-face planes: direct addressing");
  const Geometry& g = p.g;
  bool live;
  const ScNode n = sc_node<L>(g, p.y0, p.z0, g.lat_nx - 2, live, p.xcd_shift);
  if (!live) return;
  const uint32_t gi = n.gi;
  uint32_t si = gi;
  if constexpr (INDIRECT) {
    si = p.nodes[gi];
    if (si == INVALID_NODE) return;
  }
  if constexpr (GENERAL) {
    const uint32_t code = p.map[gi];
    const int kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
    if (!kind_is_wet(kind)) return;
  }
  const size_t ds = g.dist_size;
  R f[L::Q];
  // lattice 0
  if constexpr (XF && PROP == PROP_AA_ODD && !GENERAL) {
    sc_load<L, R, PROP, INDIRECT>(f, p.d_in, ds, sc_pull_node(g, n, g.lat_nx - 2, p.xrecv[0][0] != nullptr, p.xrecv[0][1] != nullptr));
  } else {
    sc_load<L, R, PROP, INDIRECT>(f, p.d_in, ds, n, p.nodes, si);
  }
  if constexpr (XF) sc_face_receive<L, R, PROP == PROP_AA_ODD>(p.xrecv[0], f, n.gx, g.lat_nx - 2, face_rows<L>(g, n.gy, n.gz));
  const R rho0 = density<L, R>(f);
  R v[3] = {(R)0, (R)0, (R)0};
  if constexpr (VOUT) {
    v[0] = p.omega[0] * momentum<L, R, 0>(f);
    v[1] = p.omega[0] * momentum<L, R, 1>(f);
    if constexpr (L::dim == 3) v[2] = p.omega[0] * momentum<L, R, 2>(f);
  }
  // lattice 1
  if constexpr (XF && PROP == PROP_AA_ODD && !GENERAL) {
    sc_load<L, R, PROP, INDIRECT>(f, (const R*)p.d_out, ds, sc_pull_node(g, n, g.lat_nx - 2, p.xrecv[1][0] != nullptr, p.xrecv[1][1] != nullptr));
  } else {
    sc_load<L, R, PROP, INDIRECT>(f, (const R*)p.d_out, ds, n, p.nodes, si);
  }
  if constexpr (XF) sc_face_receive<L, R, PROP == PROP_AA_ODD>(p.xrecv[1], f, n.gx, g.lat_nx - 2, face_rows<L>(g, n.gy, n.gz));
  const R rho1 = density<L, R>(f);
  p.rho0[gi] = rho0;
  p.rho1[gi] = rho1;
  if constexpr (XF) {
    // what the neighbours' sweeps read across the faces
    const int mrow = n.gy + 2 * g.arr_ny * n.gz;
    if (p.msend[0] && n.gx == 1) { p.msend[0][mrow] = rho0; p.msend[0][mrow + g.arr_ny] = rho1; }
    if (p.msend[1] && n.gx == g.lat_nx - 2) { p.msend[1][mrow] = rho0; p.msend[1][mrow + g.arr_ny] = rho1; }
  }
  if constexpr (VOUT) {
    v[0] = v[0] + p.omega[1] * momentum<L, R, 0>(f);
    v[1] = v[1] + p.omega[1] * momentum<L, R, 1>(f);
    if constexpr (L::dim == 3) v[2] = v[2] + p.omega[1] * momentum<L, R, 2>(f);
    // common velocity: sum_k j_k / tau_k over sum_k rho_k / tau_k (binary_shan_chen.mako:69-83)
    const R total = p.omega[0] * rho0 + p.omega[1] * rho1;
    p.vx[gi] = v[0] / total;
    p.vy[gi] = v[1] / total;
    if constexpr (L::dim == 3) p.vz[gi] = v[2] / total;
  }
}

// "ShanChenPrepareDensities" of the odd in-place step for fluid-only rows wrapped along x inside the kernel: the
// populations through sc_pull_rows() (aligned loads), everything else as sc_macro_kernel<..., VOUT = false>.
template <class L, class R>
__global__ void __launch_bounds__(1024) sc_density_pull_kernel(const ScParams<L, R> p) {
  const Geometry& g = p.g;
  const int nx = g.lat_nx - 2;
  bool live;
  const ScNode n = sc_node<L>(g, p.y0, p.z0, nx, live, p.xcd_shift);
  R fa[L::Q], fb[L::Q];
  sc_pull_rows<L, R>(fa, fb, p.d_in, (const R*)p.d_out, g.dist_size, n, nx);
  if (!live) return;
  p.rho0[n.gi] = density<L, R>(fa);
  p.rho1[n.gi] = density<L, R>(fb);
}

// ROW: one workgroup = one whole row, streaming through row_push() (aligned stores, slf_rowpush.h); every
// thread stays until the end (barrier inside), excluded nodes and idle lanes are merely inactive.
template <class L, class R, int K, int PROP, bool GENERAL, bool ROW = false, bool INDIRECT = false>
__global__ void __launch_bounds__(1024) sc_sweep_kernel(const ScParams<L, R> p) {
  static_assert(!(ROW && INDIRECT), "indirect addressing: per-node kernels only");
  const Geometry& g = p.g;
  const int nx = g.lat_nx - 2;
  bool live;
  const ScNode n = sc_node<L>(g, p.y0, p.z0, nx, live, p.xcd_shift);
  if constexpr (!ROW) {
    if (!live) return;
  }
  const uint32_t gi = n.gi;
  uint32_t si = gi;
  if constexpr (INDIRECT) {
    si = p.nodes[gi];
    if (si == INVALID_NODE) return;
  }
  int kind = NK_FLUID;
  bool active = live;
  if constexpr (GENERAL) {
    const uint32_t code = p.map[gi];
    kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
    if constexpr (!ROW) {
      if (kind_is_excluded(kind)) return;
    } else {
      active = live && !kind_is_excluded(kind);
    }
  }
  const bool wet = kind_is_wet(kind) && active;
  const size_t ds = g.dist_size;

  R f[L::Q];
  sc_load<L, R, PROP, INDIRECT>(f, p.d_in, ds, n, p.nodes, si);
  const R* own = (K == 0) ? p.rho0 : p.rho1;
  const R rho = own[gi];
  R a[3] = {(R)0, (R)0, (R)0};
  if (wet) {
    const R* const fields[2] = {p.rho0, p.rho1};
    sc_accel<L, R, 2>(fields, p.G, rho, p.potential, n, a);
    static_for<0, L::dim>([&](auto D) { a[D] = a[D] / rho; });
    if (p.has_body_force) {
      static_for<0, L::dim>([&](auto D) { a[D] = a[D] + p.accel[D]; });
    }
  }
  R v[3];
  v[0] = p.vx[gi];
  v[1] = p.vy[gi];
  v[2] = (R)0;
  if constexpr (L::dim == 3) v[2] = p.vz[gi];
  if constexpr (GENERAL) {
    if (kind == NK_FULL_BB) bounce_back<L, R>(f);
  }
  if (wet) bgk_relax_accel<L, R>(f, rho, v, p.omega[K], p.guo_pref[K], false, true, a, p.force_edm != 0);
  sc_store<L, R, PROP, GENERAL, ROW, INDIRECT>(g, f, p.d_out, ds, n, nx, live, active, p.nodes, si);
}

// Both lattices in one pass.  The force on lattice k is  - psi(rho_k) sum_j G_kj S_j  with the stencil sums
// S_j = sum_i w_i e_i psi(rho_j(x + e_i)): the two sweeps of the reference compute the same S_0, S_1 twice (and read
// rho, phi, u twice); here they are formed once, in the operation order of sc_accel(), then lattice 0 and lattice 1 are
// loaded, collided and streamed one after the other (the populations of only one lattice live in registers at a time).
#ifndef SLF_SC_FUSED_WAVES
#define SLF_SC_FUSED_WAVES 5
#endif
#ifndef SLF_SC_ROW_PULL
#define SLF_SC_ROW_PULL 1
#endif
#ifndef SLF_SC_PARK
#define SLF_SC_PARK 1
#endif
#ifndef SLF_SC_FUSEDV_WAVES
#define SLF_SC_FUSEDV_WAVES (SLF_SC_PARK ? 5 : 4)   // two sets of populations in registers: five waves spill 50-100 bytes per lane
#endif
// OWNV ("ShanChenCollideAndPropagateFusedV"): the node's densities and the common velocity are formed here, from the
// populations the sweep loads anyway, in the operation order of sc_macro_kernel -- the same bits; rho / phi are read for
// the neighbours only, the velocity arrays not at all (the pass in front of it then stores two streams instead of five).
// Both sets of populations are in registers from the start.
// PULL (with OWNV, ROW, the odd in-place step): the populations come through sc_pull_rows() -- aligned loads.
template <class L, class R, int PROP, bool GENERAL, bool ROW = false, bool OWNV = false, bool PULL = false, bool XF = false>
__global__ void __launch_bounds__(1024, (sizeof(R) == 4 && L::dim == 3 && (ROW || PROP == PROP_AA_EVEN)) ? (OWNV ? SLF_SC_FUSEDV_WAVES : SLF_SC_FUSED_WAVES) : 4)
sc_fused_kernel(const ScParams<L, R> p) {
  static_assert(!XF || (OWNV && !PULL && ROW == (PROP != PROP_AA_EVEN)),
                "x-face planes: whole-row kernels for the x-streaming steps, the sweep that forms its own moments");
  const Geometry& g = p.g;
  const int nx = g.lat_nx - 2;
  bool live;
  const ScNode n = sc_node<L>(g, p.y0, p.z0, nx, live, p.xcd_shift);
  if constexpr (!ROW) {
    if (!live) return;
  }
  const uint32_t gi = n.gi;
  int kind = NK_FLUID;
  bool active = live;
  if constexpr (GENERAL) {
    const uint32_t code = p.map[gi];
    kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
    if constexpr (!ROW) {
      if (kind_is_excluded(kind)) return;
    } else {
      active = live && !kind_is_excluded(kind);
    }
  }
  bool wet = kind_is_wet(kind) && active;
  if constexpr (OWNV && !GENERAL) {
    // fluid-only: keep `wet` a run-time value, so that the blocks it guards stay blocks -- with the guard folded away the
    // scheduler pulls the collision's temporaries over the two sets of populations and spills (96 VGPRs + 20-24 bytes)
    int one = 1;
    asm volatile("" : "+s"(one));
    wet = wet && (one != 0);
  }
  const size_t ds = g.dist_size;
  R rho[2];
  R vc[3] = {(R)0, (R)0, (R)0};
  if constexpr (!OWNV) {
    rho[0] = p.rho0[gi];
    rho[1] = p.rho1[gi];
    vc[0] = p.vx[gi];
    vc[1] = p.vy[gi];
    if constexpr (L::dim == 3) vc[2] = p.vz[gi];
  }
  // stencil sums of the fields some lattice is coupled to
  R S[2][3] = {{(R)0, (R)0, (R)0}, {(R)0, (R)0, (R)0}};
  // XF: the stencil of an edge node reaches into the neighbour subdomain -- those five values per field come from the
  // receive plane; the lane's own loads stay inside the row (the ghost column's line is never fetched)
  [[maybe_unused]] bool is_lo = false, is_hi = false;
  [[maybe_unused]] const R* mplane = nullptr;
  [[maybe_unused]] int moff[count_x_dirs<L>()];
  ScNode ns = n;
  if constexpr (XF) {
    is_lo = live && p.mrecv[0] && n.gx == 1;
    is_hi = live && p.mrecv[1] && n.gx == nx;
    if (is_lo) ns.ox.m = 0;
    if (is_hi) ns.ox.p = 0;
    const ScMacroRows mr = sc_macro_rows(g, n.gy, n.gz);
    int offp[count_x_dirs<L>()], offm[count_x_dirs<L>()];
    sc_macro_offsets<L, true>(mr, offp);
    sc_macro_offsets<L, false>(mr, offm);
    static_for<0, count_x_dirs<L>()>([&](auto K) { moff[K] = mr.row + (is_lo ? offm[K] : offp[K]); });
    mplane = is_lo ? p.mrecv[0] : p.mrecv[1];
  }
  if (wet) {
    const R* const fields[2] = {p.rho0, p.rho1};
    sc_with_potential(p.potential, [&](auto POT) {
      static_for<0, 2>([&](auto J) {
        if (p.G[J] != (R)0 || p.G2[J] != (R)0) {
          [[maybe_unused]] R ev[count_x_dirs<L>()];
          if constexpr (XF) {
            static_for<0, count_x_dirs<L>()>([&](auto K) { ev[K] = (R)0; });
            if (is_lo || is_hi) {
              static_for<0, count_x_dirs<L>()>([&](auto K) { ev[K] = mplane[moff[K] + (int)J * g.arr_ny]; });
            }
          }
          static_for<1, L::Q>([&](auto I) {
            R nb = *sc_neighbour<L, I>(fields[J], ns, true);
            if constexpr (XF && L::ex(I) > 0) nb = is_hi ? ev[x_dir_rank<L, I>()] : nb;
            if constexpr (XF && L::ex(I) < 0) nb = is_lo ? ev[x_dir_rank<L, I>()] : nb;
            const R psi = sc_psi<R, POT>(nb);
            static_for<0, L::dim>([&](auto D) {
              constexpr int e = e_comp<L>(I, D);
              if constexpr (e > 0) S[J][D] = S[J][D] + psi * Weights<L, R>::w(I);
              if constexpr (e < 0) S[J][D] = S[J][D] + psi * ((R)0 - Weights<L, R>::w(I));
            });
          });
        }
      });
    });
  }
  R fa[OWNV ? L::Q : 1], fb[OWNV ? L::Q : 1];
  if constexpr (OWNV) {
    // after the stencil sums: their 36 neighbour values and the 38 populations are not in registers together
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PULL) {
      static_assert(OWNV && ROW && PROP == PROP_AA_ODD, "the aligned pull serves the whole-row odd step");
      sc_pull_rows<L, R>(fa, fb, p.d_in, p.d_in2, ds, n, nx);
    } else if constexpr (XF && PROP == PROP_AA_ODD && !GENERAL) {
      sc_load<L, R, PROP>(fa, p.d_in, ds, sc_pull_node(g, n, nx, p.xrecv[0][0] != nullptr, p.xrecv[0][1] != nullptr));
      sc_load<L, R, PROP>(fb, p.d_in2, ds, sc_pull_node(g, n, nx, p.xrecv[1][0] != nullptr, p.xrecv[1][1] != nullptr));
    } else {
      sc_load<L, R, PROP>(fa, p.d_in, ds, n);
      sc_load<L, R, PROP>(fb, p.d_in2, ds, n);
    }
    if constexpr (XF) {
      if (live) {
        const FaceRows fr = face_rows<L>(g, n.gy, n.gz);
        sc_face_receive<L, R, PROP == PROP_AA_ODD>(p.xrecv[0], fa, n.gx, nx, fr);
        sc_face_receive<L, R, PROP == PROP_AA_ODD>(p.xrecv[1], fb, n.gx, nx, fr);
      }
    }
    rho[0] = density<L, R>(fa);
    vc[0] = p.omega[0] * momentum<L, R, 0>(fa);
    vc[1] = p.omega[0] * momentum<L, R, 1>(fa);
    if constexpr (L::dim == 3) vc[2] = p.omega[0] * momentum<L, R, 2>(fa);
    rho[1] = density<L, R>(fb);
    vc[0] = vc[0] + p.omega[1] * momentum<L, R, 0>(fb);
    vc[1] = vc[1] + p.omega[1] * momentum<L, R, 1>(fb);
    if constexpr (L::dim == 3) vc[2] = vc[2] + p.omega[1] * momentum<L, R, 2>(fb);
    const R total = p.omega[0] * rho[0] + p.omega[1] * rho[1];
    vc[0] = vc[0] / total;
    vc[1] = vc[1] / total;
    if constexpr (L::dim == 3) vc[2] = vc[2] / total;
  }
  R psi_own[2] = {rho[0], rho[1]};
  if (p.potential != 0) {
    psi_own[0] = sc_psi<R, 1>(rho[0]);
    psi_own[1] = sc_psi<R, 1>(rho[1]);
  }
  auto finish = [&](auto K, R (&f)[L::Q]) {
    R a[3] = {(R)0, (R)0, (R)0};
    if (wet) {
      const R psi_loc = psi_own[K];
      static_for<0, 2>([&](auto J) {
        const R cc = (K == 0) ? p.G[J] : p.G2[J];
        if (cc != (R)0) {
          static_for<0, L::dim>([&](auto D) { a[D] = a[D] + S[J][D] * (((R)0 - psi_loc) * cc); });
        }
      });
      static_for<0, L::dim>([&](auto D) { a[D] = a[D] / rho[K]; });
      if (K == 0 ? p.has_body_force : p.has_body_force2) {
        static_for<0, L::dim>([&](auto D) { a[D] = a[D] + (K == 0 ? p.accel[D] : p.accel2[D]); });
      }
    }
    R v[3] = {vc[0], vc[1], vc[2]};
    if constexpr (GENERAL) {
      if (kind == NK_FULL_BB) bounce_back<L, R>(f);
    }
    if (wet) bgk_relax_accel<L, R>(f, rho[K], v, p.omega[K], p.guo_pref[K], false, true, a, p.force_edm != 0);
    if constexpr (ROW && PROP != PROP_AA_EVEN && K == 1) __syncthreads();     // row_push's LDS words are still being read
    if constexpr (XF && ROW) {
      const FaceRows fr = face_rows<L>(g, n.gy, n.gz);
      row_push<L, R, GENERAL, sc_nt<L>()>(g, f, K == 0 ? p.d_out : p.d_out2, ds, n.row, n.xi, n.gx, nx, live, active, n.oy,
                                          n.oz, p.xsend[K], &fr);
    } else if constexpr (XF) {
      sc_store<L, R, PROP, GENERAL, ROW>(g, f, K == 0 ? p.d_out : p.d_out2, ds, n, nx, live, active);
      sc_face_send_own_row<L, R>(p.xsend[K], f, n.gx, nx, face_rows<L>(g, n.gy, n.gz));
    } else {
      sc_store<L, R, PROP, GENERAL, ROW>(g, f, K == 0 ? p.d_out : p.d_out2, ds, n, nx, live, active);
    }
  };
  if constexpr (OWNV) {
    // single precision D3Q19: lattice 1 waits in LDS while lattice 0 collides and streams -- registers as for one
    // lattice, a resident wave more.  80 bytes per lane, lane after lane: ONE address register with immediate offsets,
    // and 16 consecutive lanes' 16-byte accesses fall on 16 different groups of four banks (20 l mod 64)
    constexpr bool PARK = SLF_SC_PARK && sizeof(R) == 4 && L::Q == 19;
    if constexpr (PARK) {
      extern __shared__ float4 sc_park[];
      static_for<0, 5>([&](auto K4) {
        constexpr int b = 4 * K4;
        sc_park[5 * threadIdx.x + K4] = make_float4(fb[b], fb[b + 1], fb[b + 2], b + 3 < 19 ? fb[b + 3 < 19 ? b + 3 : 18] : 0.0f);
      });
      __builtin_amdgcn_sched_barrier(0);      // parked before the collision's temporaries come to life
    }
    finish(std::integral_constant<int, 0>{}, fa);
    if constexpr (PARK) {
      extern __shared__ float4 sc_park[];
      static_for<0, 5>([&](auto K4) {
        constexpr int b = 4 * K4;
        const float4 t = sc_park[5 * threadIdx.x + K4];
        fb[b] = t.x; fb[b + 1] = t.y; fb[b + 2] = t.z;
        if constexpr (b + 3 < 19) fb[b + 3] = t.w;
      });
    }
    finish(std::integral_constant<int, 1>{}, fb);
  } else {
    static_for<0, 2>([&](auto K) {
      // one lattice at a time: without the fence the scheduler starts the loads of lattice 1 under the collision of
      // lattice 0, and two sets of populations in registers cost a resident wave
      __builtin_amdgcn_sched_barrier(0);
      R f[L::Q];
      sc_load<L, R, PROP>(f, K == 0 ? p.d_in : p.d_in2, ds, n);
      finish(K, f);
    });
  }
}

// ---- single-component Shan-Chen (reference lb_single.py:242-347, lb_single_fluid.mako:129-229) ----
// PrepareMacroFields: density of every wet node
template <class L, class R, int PROP, bool GENERAL, bool INDIRECT = false, bool XF = false>
__global__ void __launch_bounds__(1024) scs_macro_kernel(const ScParams<L, R> p) {
  static_assert(!XF || !INDIRECT, "x-face planes: direct addressing");
  const Geometry& g = p.g;
  bool live;
  const ScNode n = sc_node<L>(g, p.y0, p.z0, g.lat_nx - 2, live, p.xcd_shift);
  if (!live) return;
  uint32_t si = n.gi;
  if constexpr (INDIRECT) {
    si = p.nodes[n.gi];
    if (si == INVALID_NODE) return;
  }
  if constexpr (GENERAL) {
    const uint32_t code = p.map[n.gi];
    const int kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
    if (kind_is_excluded(kind)) return;
  }
  R f[L::Q];
  if constexpr (XF && PROP == PROP_AA_ODD && !GENERAL) {
    sc_load<L, R, PROP, INDIRECT>(f, p.d_in, g.dist_size, sc_pull_node(g, n, g.lat_nx - 2, p.xrecv[0][0] != nullptr, p.xrecv[0][1] != nullptr));
  } else {
    sc_load<L, R, PROP, INDIRECT>(f, p.d_in, g.dist_size, n, p.nodes, si);
  }
  if constexpr (XF) sc_face_receive<L, R, PROP == PROP_AA_ODD>(p.xrecv[0], f, n.gx, g.lat_nx - 2, face_rows<L>(g, n.gy, n.gz));
  const R rho = density<L, R>(f);
  p.rho0[n.gi] = rho;
  if constexpr (XF) {
    const int mrow = n.gy + 2 * g.arr_ny * n.gz;      // the binary model's plane layout, field 0 only
    if (p.msend[0] && n.gx == 1) p.msend[0][mrow] = rho;
    if (p.msend[1] && n.gx == g.lat_nx - 2) p.msend[1][mrow] = rho;
  }
}

// CollideAndPropagate with the self-interaction force F = -G psi(rho(x)) sum_i w_i e_i psi(rho(x + e_i))
template <class L, class R, int PROP, bool GENERAL, bool ROW = false, bool INDIRECT = false, bool XF = false>
__global__ void __launch_bounds__(1024) scs_sweep_kernel(const ScParams<L, R> p) {
  static_assert(!(ROW && INDIRECT), "indirect addressing: per-node kernels only");
  static_assert(!XF || (!INDIRECT && ROW == (PROP != PROP_AA_EVEN)), "x-face planes: whole-row kernels for the x-streaming steps");
  const Geometry& g = p.g;
  const int nx = g.lat_nx - 2;
  bool live;
  const ScNode n = sc_node<L>(g, p.y0, p.z0, nx, live, p.xcd_shift);
  if constexpr (!ROW) {
    if (!live) return;
  }
  const uint32_t gi = n.gi;
  uint32_t si = gi;
  if constexpr (INDIRECT) {
    si = p.nodes[gi];
    if (si == INVALID_NODE) return;
  }
  int kind = NK_FLUID;
  bool active = live;
  if constexpr (GENERAL) {
    const uint32_t code = p.map[gi];
    kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
    if constexpr (!ROW) {
      if (kind_is_excluded(kind)) return;
    } else {
      active = live && !kind_is_excluded(kind);
    }
  }
  const bool wet = kind_is_wet(kind) && active;
  const size_t ds = g.dist_size;
  R f[L::Q];
  if constexpr (XF && PROP == PROP_AA_ODD && !GENERAL) {
    sc_load<L, R, PROP, INDIRECT>(f, p.d_in, ds, sc_pull_node(g, n, nx, p.xrecv[0][0] != nullptr, p.xrecv[0][1] != nullptr));
  } else {
    sc_load<L, R, PROP, INDIRECT>(f, p.d_in, ds, n, p.nodes, si);
  }
  if constexpr (XF) {
    if (live) sc_face_receive<L, R, PROP == PROP_AA_ODD>(p.xrecv[0], f, n.gx, nx, face_rows<L>(g, n.gy, n.gz));
  }
  R rho, v[3];
  macro_standard<L, R>(f, false, rho, v);
  R a[3] = {(R)0, (R)0, (R)0};
  if (wet) {
    const R* const fields[2] = {p.rho0, p.rho0};
    if constexpr (XF) {
      ScNode ns;
      const ScEdge<L, R> edge = sc_edge<L, R>(p, n, nx, live, ns);
      sc_accel_edge<L, R>(p.rho0, p.G[0], rho, p.potential, ns, edge, a);
    } else {
      sc_accel<L, R, 1>(fields, p.G, rho, p.potential, n, a);
    }
    static_for<0, L::dim>([&](auto D) { a[D] = a[D] / rho; });
    if (p.has_body_force) {
      static_for<0, L::dim>([&](auto D) { a[D] = a[D] + p.accel[D]; });
    }
  }
  if constexpr (GENERAL) {
    if (kind == NK_FULL_BB) bounce_back<L, R>(f);
  }
  if (wet) bgk_relax_accel<L, R>(f, rho, v, p.omega[0], p.guo_pref[0], false, true, a, p.force_edm != 0);
  if ((p.options & 1u) && wet) {
    // the density field itself is written by PrepareMacroFields; v is the force-shifted output velocity
    p.vx[gi] = v[0];
    p.vy[gi] = v[1];
    if constexpr (L::dim == 3) p.vz[gi] = v[2];
  }
  if constexpr (XF && ROW) {
    const FaceRows fr = face_rows<L>(g, n.gy, n.gz);
    row_push<L, R, GENERAL, sc_nt<L>()>(g, f, p.d_out, ds, n.row, n.xi, n.gx, nx, live, active, n.oy, n.oz, p.xsend[0], &fr);
  } else {
    sc_store<L, R, PROP, GENERAL, ROW, INDIRECT>(g, f, p.d_out, ds, n, nx, live, active, p.nodes, si);
    if constexpr (XF) sc_face_send_own_row<L, R>(p.xsend[0], f, n.gx, nx, face_rows<L>(g, n.gy, n.gz));
  }
}

// f1 = feq(rho, v), f2 = feq(phi, v) on every node (no type test)
template <class L, class R>
__global__ void __launch_bounds__(256) sc_init_kernel(R* d1, R* d2, const R* __restrict__ irho, const R* __restrict__ iphi,
                                                     const R* __restrict__ ivx, const R* __restrict__ ivy,
                                                     const R* __restrict__ ivz, Geometry g, const uint32_t* __restrict__ nodes) {
  const int gx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int gy = (int)blockIdx.y;
  const int gz = (int)blockIdx.z;
  if (gx > g.lat_nx - 1) return;
  const uint32_t gi = (uint32_t)gx + (uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz;
  uint32_t si = gi;       // indirect addressing: the node's slot, if it is active
  if (nodes) {
    si = nodes[gi];
    if (si == INVALID_NODE) return;
  }
  R v[3];
  v[0] = ivx[gi];
  v[1] = ivy[gi];
  v[2] = (R)0;
  if constexpr (L::dim == 3) v[2] = ivz[gi];
  const R u15 = usq15<L, R>(v);
  const R r0 = irho[gi], r1 = iphi[gi];
  static_for<0, L::Q>([&](auto I) {
    (d1 + (size_t)g.dist_size * (size_t)I)[si] = feq<L, R, I>(r0, r0, v, u15);
    (d2 + (size_t)g.dist_size * (size_t)I)[si] = feq<L, R, I>(r1, r1, v, u15);
  });
}

// ---------------------------------------------------------------------------
template <class L, class R>
static ScParams<L, R> make_sc(const Geometry& g, const Physics& ph, const ShanChen& sc, const SweepArgs& a, int grid_idx,
                              int y0, int z0) {
  ScParams<L, R> p;
  p.nodes = (const uint32_t*)a.nodes;
  p.map = (const uint32_t*)a.map;
  p.d_in = (const R*)a.dist_in;
  p.d_out = (R*)a.dist_out;
  p.rho0 = (R*)a.rho;
  p.rho1 = (R*)a.phi;
  p.vx = (R*)a.v[0];
  p.vy = (R*)a.v[1];
  p.vz = (R*)a.v[2];
  p.options = a.options;
  p.y0 = y0;
  p.z0 = z0;
  p.g = g;
  p.omega[0] = (R)(1.0 / ph.tau);
  p.omega[1] = (R)(1.0 / sc.tau_phi);
  p.guo_pref[0] = (R)(3.0 * (1.0 - 0.5 / ph.tau));
  p.guo_pref[1] = (R)(3.0 * (1.0 - 0.5 / sc.tau_phi));
  p.G[0] = (R)sc.G[2 * grid_idx + 0];
  p.G[1] = (R)sc.G[2 * grid_idx + 1];
  p.force_edm = ph.force_edm;
  // body forces act per lattice (reference add_body_force(force, grid=k), relaxation_common.mako:9-36)
  p.has_body_force = 0;
  for (int d = 0; d < 3; d++) {
    p.accel[d] = (R)(grid_idx == 0 ? ph.accel[d] : sc.accel1[d]);
    if (p.accel[d] != (R)0) p.has_body_force = 1;
  }
  p.potential = sc.potential;
  p.d_in2 = (const R*)a.dist_in2;
  p.d_out2 = (R*)a.dist_out2;
  p.G2[0] = (R)sc.G[2];
  p.G2[1] = (R)sc.G[3];
  p.xcd_shift = 0;
  for (int f = 0; f < 2; f++) {
    p.xsend[0][f] = (R*)a.xsend[f];
    p.xsend[1][f] = (R*)a.xsend2[f];
    p.xrecv[0][f] = (const R*)a.xrecv[f];
    p.xrecv[1][f] = (const R*)a.xrecv2[f];
    p.msend[f] = (R*)a.msend[f];
    p.mrecv[f] = (const R*)a.mrecv[f];
  }
  p.has_body_force2 = 0;
  for (int d = 0; d < 3; d++) {
    p.accel2[d] = (R)sc.accel1[d];
    if (p.accel2[d] != (R)0) p.has_body_force2 = 1;
  }
  return p;
}

static inline bool sc_xface_in_use(const SweepArgs& a) {
  for (int f = 0; f < 2; f++) {
    if (a.xsend[f] || a.xrecv[f] || a.xsend2[f] || a.xrecv2[f] || a.msend[f] || a.mrecv[f]) return true;
  }
  return false;
}

// rows sc_pull_rows() serves: wrapped along x inside the kernel, the whole row in one workgroup, whole-row kernels on
static inline bool sc_row_pull_ok(const Geometry& g) {
  return SLF_SC_ROW_PULL && g.wrap[0] && g.lat_nx - 2 <= 1024 && (g.variant & 8) && !g.indirect;
}

template <class L, class R>
static hipError_t sc_macro2(Prop prop, bool general, const Geometry& g, const Physics& ph, const ShanChen& sc,
                            const SweepArgs& a, int y0, int y1, int z0, int z1, hipStream_t s) {
  ScParams<L, R> p = make_sc<L, R>(g, ph, sc, a, 0, y0, z0);
  const int nx = g.lat_nx - 2;
  int bx = ((nx + 63) / 64) * 64;
  if (bx > 1024) bx = 256;
  dim3 block(bx, 1, 1);
  dim3 grid((nx + bx - 1) / bx, y1 - y0, L::dim == 3 ? z1 - z0 : 1);
  if (grid.y == 0 || grid.z == 0) return hipSuccess;
  // the densities-only form: indirect addressing keeps the reference's pass
  const bool vout = !a.sc_local_velocity || (a.options & 1u) || g.indirect;
  if (sc_xface_in_use(a)) {
    // connected x faces through planes: edge lanes take the entering populations from the receive planes and store their
    // densities for the neighbours (the module was checked when the planes were set: slf_module_set_xface_planes)
    if constexpr (L::dim == 3) {
      if (g.indirect) return hipErrorInvalidValue;
      p.xcd_shift = xcd_shift_for(grid.y, grid.x);        // 32 consecutive rows write one line of a plane: one XCD
#define SLF_SCM_XF(P, G)                                                                                            \
  do {                                                                                                               \
    if (vout) hipLaunchKernelGGL((sc_macro_kernel<L, R, P, G, false, true, true>), grid, block, 0, s, p);            \
    else hipLaunchKernelGGL((sc_macro_kernel<L, R, P, G, false, false, true>), grid, block, 0, s, p);                \
  } while (0)
      if (prop == PROP_AA_ODD) {
        if (general) SLF_SCM_XF(PROP_AA_ODD, true); else SLF_SCM_XF(PROP_AA_ODD, false);
      } else {          // two-copy and the even in-place step: the node's own slots
        if (general) SLF_SCM_XF(PROP_AB, true); else SLF_SCM_XF(PROP_AB, false);
      }
#undef SLF_SCM_XF
      return hipGetLastError();
    }
    return hipErrorInvalidValue;
  }
  if constexpr (L::dim == 3) {
    if (!vout && !general && prop == PROP_AA_ODD && sc_row_pull_ok(g)) {
      hipLaunchKernelGGL((sc_density_pull_kernel<L, R>), grid, block, 0, s, p);
      return hipGetLastError();
    }
  }
#define SLF_SCM(P)                                                                        \
  do {                                                                                    \
    if (g.indirect) hipLaunchKernelGGL((sc_macro_kernel<L, R, P, true, true>), grid, block, 0, s, p); \
    else if (general && vout) hipLaunchKernelGGL((sc_macro_kernel<L, R, P, true>), grid, block, 0, s, p); \
    else if (general) hipLaunchKernelGGL((sc_macro_kernel<L, R, P, true, false, false>), grid, block, 0, s, p); \
    else if (vout) hipLaunchKernelGGL((sc_macro_kernel<L, R, P, false>), grid, block, 0, s, p);        \
    else hipLaunchKernelGGL((sc_macro_kernel<L, R, P, false, false, false>), grid, block, 0, s, p);        \
  } while (0)
  if (prop == PROP_AA_ODD) SLF_SCM(PROP_AA_ODD);
  else SLF_SCM(PROP_AB);
#undef SLF_SCM
  return hipGetLastError();
}

template <class L, class R, int K>
static hipError_t sc_sweep3(Prop prop, bool general, bool row, const ScParams<L, R>& p, dim3 grid, dim3 block,
                            hipStream_t s) {
  if (p.g.indirect) {      // active-node slots: per-node kernels with translated neighbours
    if (prop == PROP_AB) hipLaunchKernelGGL((sc_sweep_kernel<L, R, K, PROP_AB, true, false, true>), grid, block, 0, s, p);
    else if (prop == PROP_AA_EVEN) hipLaunchKernelGGL((sc_sweep_kernel<L, R, K, PROP_AA_EVEN, true, false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((sc_sweep_kernel<L, R, K, PROP_AA_ODD, true, false, true>), grid, block, 0, s, p);
    return hipGetLastError();
  }
#define SLF_SCS(P)                                                                           \
  do {                                                                                       \
    if (general) hipLaunchKernelGGL((sc_sweep_kernel<L, R, K, P, true>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((sc_sweep_kernel<L, R, K, P, false>), grid, block, 0, s, p);        \
  } while (0)
#define SLF_SCS_ROW(P)                                                                             \
  do {                                                                                               \
    if (general) hipLaunchKernelGGL((sc_sweep_kernel<L, R, K, P, true, true>), grid, block, 0, s, p);   \
    else hipLaunchKernelGGL((sc_sweep_kernel<L, R, K, P, false, true>), grid, block, 0, s, p);          \
  } while (0)
  if constexpr (L::dim == 3) {
    if (row && prop == PROP_AB) { SLF_SCS_ROW(PROP_AB); return hipGetLastError(); }
    if (row && prop == PROP_AA_ODD) { SLF_SCS_ROW(PROP_AA_ODD); return hipGetLastError(); }
  }
  if (prop == PROP_AB) SLF_SCS(PROP_AB);
  else if (prop == PROP_AA_EVEN) SLF_SCS(PROP_AA_EVEN);
  else SLF_SCS(PROP_AA_ODD);
#undef SLF_SCS
#undef SLF_SCS_ROW
  return hipGetLastError();
}

template <class L, class R>
static hipError_t sc_sweep2(int grid_idx, Prop prop, bool general, const Geometry& g, const Physics& ph,
                            const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x,
                            hipStream_t s) {
  if (sc_xface_in_use(a)) return hipErrorInvalidValue;       // planes: the fused sweep only
  ScParams<L, R> p = make_sc<L, R>(g, ph, sc, a, grid_idx, y0, z0);
  const int nx = g.lat_nx - 2;
  // whole-row workgroups + aligned stores for the x-streaming steps in 3-D (as slf_row.hip)
  const bool row = L::dim == 3 && (g.variant & 8) && prop != PROP_AA_EVEN && !g.indirect;
  if (row) block_x = row_block_x(nx);
  dim3 block(block_x, 1, 1);
  dim3 grid((nx + block_x - 1) / block_x, y1 - y0, L::dim == 3 ? z1 - z0 : 1);
  if (grid.y == 0 || grid.z == 0) return hipSuccess;
  p.xcd_shift = xcd_shift_for(grid.y, grid.x);      // the force stencil's rho / phi rows: neighbouring rows on one XCD
  if (grid_idx == 0) return sc_sweep3<L, R, 0>(prop, general, row, p, grid, block, s);
  return sc_sweep3<L, R, 1>(prop, general, row, p, grid, block, s);
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

hipError_t launch_sc_macro(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                           const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, hipStream_t s) {
  SLF_DISPATCH_LR(sel, return (sc_macro2<L, R>(prop, sel.general, g, ph, sc, a, y0, y1, z0, z1, s)));
  return hipErrorInvalidValue;
}

hipError_t launch_sc_sweep(const KernelSelector& sel, int grid_idx, Prop prop, const Geometry& g, const Physics& ph,
                           const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x,
                           hipStream_t s) {
  SLF_DISPATCH_LR(sel, return (sc_sweep2<L, R>(grid_idx, prop, sel.general, g, ph, sc, a, y0, y1, z0, z1, block_x, s)));
  return hipErrorInvalidValue;
}

template <class L, class R>
static hipError_t sc_fused2(Prop prop, bool general, const Geometry& g, const Physics& ph, const ShanChen& sc,
                            const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x, hipStream_t s) {
  ScParams<L, R> p = make_sc<L, R>(g, ph, sc, a, 0, y0, z0);
  const int nx = g.lat_nx - 2;
  const bool row = L::dim == 3 && (g.variant & 8) && prop != PROP_AA_EVEN;
  if (row) block_x = row_block_x(nx);
  dim3 block(block_x, 1, 1);
  dim3 grid((nx + block_x - 1) / block_x, y1 - y0, L::dim == 3 ? z1 - z0 : 1);
  if (grid.y == 0 || grid.z == 0) return hipSuccess;
  p.xcd_shift = xcd_shift_for(grid.y, grid.x);      // the force stencil's rho / phi rows: neighbouring rows on one XCD
  const size_t park = (SLF_SC_PARK && sizeof(R) == 4 && L::Q == 19) ? (size_t)block_x * 5 * 16 : 0;   // lattice 1 parked in LDS
#define SLF_SCF(P, ROW)                                                                            \
  do {                                                                                             \
    if (a.sc_local_velocity) {                                                                     \
      if (general) hipLaunchKernelGGL((sc_fused_kernel<L, R, P, true, ROW, true>), grid, block, park, s, p);   \
      else hipLaunchKernelGGL((sc_fused_kernel<L, R, P, false, ROW, true>), grid, block, park, s, p);          \
    } else {                                                                                       \
      if (general) hipLaunchKernelGGL((sc_fused_kernel<L, R, P, true, ROW>), grid, block, 0, s, p);   \
      else hipLaunchKernelGGL((sc_fused_kernel<L, R, P, false, ROW>), grid, block, 0, s, p);          \
    }                                                                                              \
  } while (0)
  if (sc_xface_in_use(a)) {
    if constexpr (L::dim == 3) {
      if (!a.sc_local_velocity) return hipErrorInvalidValue;
#define SLF_SCF_XF(P, ROW)                                                                                                   \
  do {                                                                                                                        \
    if (general) hipLaunchKernelGGL((sc_fused_kernel<L, R, P, true, ROW, true, false, true>), grid, block, park, s, p);       \
    else hipLaunchKernelGGL((sc_fused_kernel<L, R, P, false, ROW, true, false, true>), grid, block, park, s, p);              \
  } while (0)
      if (prop == PROP_AA_EVEN) {
        SLF_SCF_XF(PROP_AA_EVEN, false);
        return hipGetLastError();
      }
      if (!row || grid.x != 1) return hipErrorInvalidValue;
      if (prop == PROP_AB) SLF_SCF_XF(PROP_AB, true);
      else SLF_SCF_XF(PROP_AA_ODD, true);
#undef SLF_SCF_XF
      return hipGetLastError();
    }
    return hipErrorInvalidValue;
  }
  if constexpr (L::dim == 3) {
    if (row && prop == PROP_AB) { SLF_SCF(PROP_AB, true); return hipGetLastError(); }
    if (row && prop == PROP_AA_ODD && a.sc_local_velocity && sc_row_pull_ok(g) && grid.x == 1) {
      if (general) hipLaunchKernelGGL((sc_fused_kernel<L, R, PROP_AA_ODD, true, true, true, true>), grid, block, park, s, p);
      else hipLaunchKernelGGL((sc_fused_kernel<L, R, PROP_AA_ODD, false, true, true, true>), grid, block, park, s, p);
      return hipGetLastError();
    }
    if (row && prop == PROP_AA_ODD) { SLF_SCF(PROP_AA_ODD, true); return hipGetLastError(); }
  }
  if (prop == PROP_AB) SLF_SCF(PROP_AB, false);
  else if (prop == PROP_AA_EVEN) SLF_SCF(PROP_AA_EVEN, false);
  else SLF_SCF(PROP_AA_ODD, false);
#undef SLF_SCF
  return hipGetLastError();
}

hipError_t launch_sc_fused(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph, const ShanChen& sc,
                           const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x, hipStream_t s) {
  SLF_DISPATCH_LR(sel, return (sc_fused2<L, R>(prop, sel.general, g, ph, sc, a, y0, y1, z0, z1, block_x, s)));
  return hipErrorInvalidValue;
}

template <class L, class R>
static hipError_t scs_launch2(bool macro, Prop prop, bool general, const Geometry& g, const Physics& ph,
                              const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x,
                              hipStream_t s) {
  ScParams<L, R> p = make_sc<L, R>(g, ph, sc, a, 0, y0, z0);
  p.G[0] = (R)sc.G[0];
  p.G[1] = (R)0;
  const int nx = g.lat_nx - 2;
  const bool row = !macro && L::dim == 3 && (g.variant & 8) && prop != PROP_AA_EVEN && !g.indirect;
  if (row) block_x = row_block_x(nx);
  dim3 block(block_x, 1, 1);
  dim3 grid((nx + block_x - 1) / block_x, y1 - y0, L::dim == 3 ? z1 - z0 : 1);
  if (grid.y == 0 || grid.z == 0) return hipSuccess;
  p.xcd_shift = xcd_shift_for(grid.y, grid.x);      // the force stencil's rho / phi rows: neighbouring rows on one XCD
  if (sc_xface_in_use(a)) {
    // connected x faces through planes (slf_module_set_xface_planes: sets 0 and 2 of the binary model's three)
    if constexpr (L::dim == 3) {
      if (g.indirect) return hipErrorInvalidValue;
#define SLF_SCS_XF(KERN, P, ...)                                                                             \
  do {                                                                                                        \
    if (general) hipLaunchKernelGGL((KERN<L, R, P, true, __VA_ARGS__>), grid, block, 0, s, p);                \
    else hipLaunchKernelGGL((KERN<L, R, P, false, __VA_ARGS__>), grid, block, 0, s, p);                       \
  } while (0)
      if (macro) {
        if (prop == PROP_AA_ODD) SLF_SCS_XF(scs_macro_kernel, PROP_AA_ODD, false, true);
        else SLF_SCS_XF(scs_macro_kernel, PROP_AB, false, true);
        return hipGetLastError();
      }
      if (prop == PROP_AA_EVEN) {
        SLF_SCS_XF(scs_sweep_kernel, PROP_AA_EVEN, false, false, true);
        return hipGetLastError();
      }
      if (!row || grid.x != 1) return hipErrorInvalidValue;
      if (prop == PROP_AB) SLF_SCS_XF(scs_sweep_kernel, PROP_AB, true, false, true);
      else SLF_SCS_XF(scs_sweep_kernel, PROP_AA_ODD, true, false, true);
#undef SLF_SCS_XF
      return hipGetLastError();
    }
    return hipErrorInvalidValue;
  }
  if (g.indirect) {      // active-node slots: per-node kernels with translated neighbours (the node map is always read)
    if (macro) {
      if (prop == PROP_AA_ODD) hipLaunchKernelGGL((scs_macro_kernel<L, R, PROP_AA_ODD, true, true>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((scs_macro_kernel<L, R, PROP_AB, true, true>), grid, block, 0, s, p);
    } else {
      if (prop == PROP_AB) hipLaunchKernelGGL((scs_sweep_kernel<L, R, PROP_AB, true, false, true>), grid, block, 0, s, p);
      else if (prop == PROP_AA_EVEN) hipLaunchKernelGGL((scs_sweep_kernel<L, R, PROP_AA_EVEN, true, false, true>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((scs_sweep_kernel<L, R, PROP_AA_ODD, true, false, true>), grid, block, 0, s, p);
    }
    return hipGetLastError();
  }
  if constexpr (L::dim == 3) {
    if (row) {
      if (prop == PROP_AB) {
        if (general) hipLaunchKernelGGL((scs_sweep_kernel<L, R, PROP_AB, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((scs_sweep_kernel<L, R, PROP_AB, false, true>), grid, block, 0, s, p);
      } else {
        if (general) hipLaunchKernelGGL((scs_sweep_kernel<L, R, PROP_AA_ODD, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((scs_sweep_kernel<L, R, PROP_AA_ODD, false, true>), grid, block, 0, s, p);
      }
      return hipGetLastError();
    }
  }
#define SLF_SCS1(KERN, P)                                                       \
  do {                                                                          \
    if (general) hipLaunchKernelGGL((KERN<L, R, P, true>), grid, block, 0, s, p);  \
    else hipLaunchKernelGGL((KERN<L, R, P, false>), grid, block, 0, s, p);         \
  } while (0)
  if (macro) {
    if (prop == PROP_AA_ODD) SLF_SCS1(scs_macro_kernel, PROP_AA_ODD);
    else SLF_SCS1(scs_macro_kernel, PROP_AB);
  } else {
    if (prop == PROP_AB) SLF_SCS1(scs_sweep_kernel, PROP_AB);
    else if (prop == PROP_AA_EVEN) SLF_SCS1(scs_sweep_kernel, PROP_AA_EVEN);
    else SLF_SCS1(scs_sweep_kernel, PROP_AA_ODD);
  }
#undef SLF_SCS1
  return hipGetLastError();
}

hipError_t launch_scs_macro(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                            const ShanChen& sc, const SweepArgs& a, hipStream_t s) {
  const int z0 = g.dim == 3 ? 1 : 0, z1 = g.dim == 3 ? g.lat_nz - 1 : 1;
  int bx = ((g.lat_nx - 2 + 63) / 64) * 64;
  if (bx > 1024) bx = 256;
  SLF_DISPATCH_LR(sel, return (scs_launch2<L, R>(true, prop, sel.general, g, ph, sc, a, 1, g.lat_ny - 1, z0, z1, bx, s)));
  return hipErrorInvalidValue;
}

hipError_t launch_scs_sweep(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                            const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x,
                            hipStream_t s) {
  SLF_DISPATCH_LR(sel, return (scs_launch2<L, R>(false, prop, sel.general, g, ph, sc, a, y0, y1, z0, z1, block_x, s)));
  return hipErrorInvalidValue;
}

template <class L, class R>
static hipError_t sc_init2(const Geometry& g, void* d1, void* d2, const void* rho, const void* phi,
                           const void* const v[3], const void* nodes, hipStream_t s) {
  dim3 block(256, 1, 1);
  dim3 grid((g.lat_nx + 255) / 256, g.lat_ny, g.lat_nz);
  hipLaunchKernelGGL((sc_init_kernel<L, R>), grid, block, 0, s, (R*)d1, (R*)d2, (const R*)rho, (const R*)phi,
                     (const R*)v[0], (const R*)v[1], (const R*)v[2], g, (const uint32_t*)nodes);
  return hipGetLastError();
}

hipError_t launch_sc_init(const KernelSelector& sel, const Geometry& g, const Physics& ph, void* dist1, void* dist2,
                          const void* rho, const void* phi, const void* const v[3], const void* nodes, hipStream_t s) {
  SLF_DISPATCH_LR(sel, return (sc_init2<L, R>(g, dist1, dist2, rho, phi, v, nodes, s)));
  return hipErrorInvalidValue;
}

}  // namespace slf
