// Host-visible launch interface of the gfx950 kernels (C++ linkage, used only by
// slf_api.hip).  See slf_kernels.hip for the kernels themselves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slf {

enum Prop : int { PROP_AB = 0, PROP_AA_EVEN = 1, PROP_AA_ODD = 2 };

// Geometry + decode information shared by all kernels of a module.
struct Geometry {
  int dim;
  int lat_nx, lat_ny, lat_nz;  // incl. ghosts (lat_nz == 1 in 2-D)
  int arr_nx, arr_ny, arr_nz;
  int arr_nxy;                 // arr_nx * arr_ny
  uint32_t dist_size;          // stride between direction arrays (>= arr_nx * arr_ny * arr_nz)
  int wrap[3];                 // in-kernel periodic wrap per axis
  int axis_mode[3];            // 0: not locally periodic, 1: ghost-layer PBC kernels, 2: wrapped in-sweep
  // node code decoding
  uint32_t type_mask;
  uint32_t param_shift;        // = bits of the type field
  uint32_t param_mask;
  uint32_t orient_shift;
  unsigned long long type_lut; // 4 bits per dense type id -> NodeKind
  int use_link_tags;
  int variant;                 // tuned-kernel selection bits (SLF_VARIANT), see slf_fast.hip
  int indirect;                // distributions hold active nodes only, addressed through SweepArgs::nodes
  // Distribution addressing: element (q, x, y, z) lives at q * dq + x + dsy * y + dsz * z.
  //   direction-major (reference layout, kernel_common.mako:459-461):  dq = dist_size, dsy = arr_nx, dsz = arr_nxy
  //   row-interleaved [z][y][q][x]:                                     dq = arr_nx, dsy = Q arr_nx, dsz = Q arr_nx arr_ny
  unsigned long long dq;
  long long dsy, dsz;
  int layout;                  // 0 direction-major, 1 row-interleaved
  int row_order;               // workgroup -> row mapping of the whole-row kernels (SLF_ROW_ORDER), see row_of_block()
  int lds_pad;                 // extra dynamic LDS bytes per workgroup (SLF_LDS_PAD): occupancy throttle, experiments
  // What the module's node-type table contains (decided once, at module creation): 0 = nothing beyond fluid, ghost,
  // unused, propagation-only and full-way bounce-back nodes; 1 = boundary-condition nodes; 2 = also the outflow
  // nodes that read neighbouring nodes (two-copy pattern).  The f32 whole-row kernels are instantiated per level: the
  // code of conditions a simulation does not use costs registers (52 instead of 78-80 VGPRs at level 0) and, for
  // the outflow nodes, spills in the hot path of the two-copy kernel.
  int bc_level;
};

struct Physics {
  double tau, visc;
  double accel[3];
  double mrt_rates[27];
  int incompressible;
  int has_force;
  int relaxation_enabled;
  int force_edm;   // body / Shan-Chen forces by the exact difference method instead of Guo's
};

struct ShanChen {
  int enabled;      // 0 off, 1 binary mixture, 2 single component
  double tau_phi;
  double G[4];      // G11 G12 G21 G22
  int potential;    // 0 linear, 1 classic
  double accel1[3]; // body-force acceleration on lattice 1 (Physics::accel acts on lattice 0)
};

struct SweepArgs {
  const void* nodes;   // indirect addressing: dense index -> slot table (uint32), else NULL
  const void* map;
  void* dist_in;
  void* dist_out;
  void* rho;
  void* phi;        // second density field (binary Shan-Chen), else unused
  void* v[3];
  const void* node_params;
  void* status;        // device uint32[4]: invalid-value flag + position
  uint32_t options;
  // x-face buffers (slf_module_set_xface_buffers): [0] low face (x = 1 side), [1] high face; NULL = not used
  void* xsend[2];
  const void* xrecv[2];
};

// One module = one (lattice, model, precision, access pattern) specialisation.
struct KernelSelector {
  int lattice, model, precision;
  bool general;  // reads the node map / handles BC nodes
};

hipError_t launch_sweep(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                        const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x, hipStream_t s);

hipError_t launch_init(const KernelSelector& sel, const Geometry& g, const Physics& ph, void* dist,
                       const void* rho, const void* const v[3], const void* nodes, hipStream_t s);

hipError_t launch_pbc(const KernelSelector& sel, const Geometry& g, void* dist, int axis, bool with_swap,
                      hipStream_t s);

hipError_t launch_macro_pbc(const KernelSelector& sel, const Geometry& g, void* field, int axis, hipStream_t s);

hipError_t launch_sparse(const KernelSelector& sel, bool collect, const unsigned long long* idx, void* dist, void* buffer,
                         int n, hipStream_t s);

hipError_t launch_box(const KernelSelector& sel, const Geometry& g, bool collect, void* dist, void* buffer,
                      unsigned int dirs, unsigned long long base, long long col_stride, int ncols, long long row_stride,
                      int nrows, hipStream_t s);

hipError_t launch_macro(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                        const SweepArgs& a, hipStream_t s);

// ---- binary Shan-Chen (slf_sc.hip) ----
// macro pass: a.dist_in = lattice 0, a.dist_out = lattice 1 (both read), writes rho, phi, v
hipError_t launch_sc_macro(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                           const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, hipStream_t s);
// collide-and-stream of lattice `grid_idx`: reads rho, phi, v
hipError_t launch_sc_sweep(const KernelSelector& sel, int grid_idx, Prop prop, const Geometry& g, const Physics& ph,
                           const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x,
                           hipStream_t s);
// single-component Shan-Chen: density pass (a.dist_in -> a.rho) and the sweep with the force
hipError_t launch_scs_macro(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                            const ShanChen& sc, const SweepArgs& a, hipStream_t s);
hipError_t launch_scs_sweep(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                            const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x,
                            hipStream_t s);
hipError_t launch_sc_init(const KernelSelector& sel, const Geometry& g, const Physics& ph, void* dist1, void* dist2,
                          const void* rho, const void* phi, const void* const v[3], hipStream_t s);

}  // namespace slf
