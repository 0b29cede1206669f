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
  // What the module's node-type table contains (decided once, at module creation): 0 = nothing beyond fluid, ghost,
  // unused, propagation-only and full-way bounce-back nodes; 1 = boundary-condition nodes; 2 = also the outflow
  // nodes that read neighbouring nodes (two-copy pattern), the do-nothing nodes of the in-place pattern and the full-slip
  // nodes.  The f32 whole-row kernels are instantiated per level: the
  // code of conditions a simulation does not use costs registers (52 instead of 78-80 VGPRs at level 0) and, for
  // the outflow nodes, spills in the hot path of the two-copy kernel.
  int bc_level;
  // bit 0 / 1: nothing ever reads the ghost column x = 0 / x = nx + 1 (the face is neither periodic through the
  // ghost-layer kernels nor connected to another subdomain): the whole-row kernels do not store into it -- five
  // partial-line writes per row and face that HBM turns into read-modify-writes (slf_module_set_x_ghost_unused)
  int x_ghost_unused;
};

struct Physics {
  double tau, visc;
  double accel[3];
  double mrt_rates[27];
  int incompressible;
  int has_force;
  int relaxation_enabled;
  int force_edm;   // body / Shan-Chen forces by the exact difference method instead of Guo's
  // --regularized / --subgrid=les-smagorinsky (slf_module_desc; relaxation_common.mako:166-237): per-node kernels only
  int regularized, subgrid;
  double smagorinsky_const;
};

struct ShanChen {
  int enabled;      // 0 off, 1 binary mixture, 2 single component
  double tau_phi;
  double G[4];      // G11 G12 G21 G22
  int potential;    // 0 linear, 1 classic
  double accel1[3]; // body-force acceleration on lattice 1 (Physics::accel acts on lattice 0)
};

struct RowClasses;
struct SlotTable;

struct SweepArgs {
  const void* nodes;   // indirect addressing: dense index -> slot table (uint32), else NULL
  const void* map;
  void* dist_in;
  void* dist_out;
  void* dist_in2;   // second lattice (fused binary Shan-Chen sweep), else unused
  void* dist_out2;
  void* rho;
  void* phi;        // second density field (binary Shan-Chen), else unused
  void* v[3];
  const void* node_params;
  void* status;        // device uint32[4]: invalid-value flag + position
  uint32_t options;
  // binary Shan-Chen: the fused sweep forms densities and velocity of its own node itself (kernels
  // ShanChenPrepareDensities / ShanChenCollideAndPropagateFusedV); the pass in front stores rho, phi and -- only when
  // options bit 0 asks for output -- the velocity
  bool sc_local_velocity;
  // x-face buffers (slf_module_set_xface_buffers): [0] low face (x = 1 side), [1] high face; NULL = not used
  void* xsend[2];
  const void* xrecv[2];
  // indirect addressing: which node owns which slot (SlotTable), NULL = not built: one thread per DENSE node
  const SlotTable* slots;
  // row classes of the node map `map` (slf_module_classify_rows), NULL = not classified; see RowClasses
  const RowClasses* rows;
  // binary Shan-Chen over connected x faces (slf_module_set_xface_planes): xsend / xrecv above serve lattice 0, these
  // lattice 1 and the densities rho, phi ([z][field][y] planes); NULL = not used
  void* xsend2[2];
  const void* xrecv2[2];
  void* msend[2];
  const void* mrecv[2];
};

// What slf_module_classify_rows() found in a node map, per 64-node x-segment (= one wavefront of a whole-row
// workgroup) and per (y, z) row.  Segment class 0: every node of the segment is a plain fluid node -- the wave
// does not read the map at all (4 of the 156 bytes per update) and runs the straight-line fluid body; 1: other
// nodes that need no boundary-condition code (ghost, unused, propagation-only, full-way bounce-back); 2: nodes
// with boundary conditions.  Rows whose worst segment is class 2 are listed in bc_rows and swept by the kernel
// instantiated for the module's Geometry::bc_level; all other rows by the level-0 instantiation with its 8
// resident waves per SIMD -- a lid-driven cavity has boundary-condition nodes in 0.2 % of its rows.
// Indirect addressing, the other way round: slot -> node.  Built once per `nodes` table (slf_api.hip), so that the sweep
// can be launched over the SLOTS -- every lane of every wave an active node, and consecutive lanes consecutive slots of
// every array -- instead of over the dense box, where a packed bed leaves 70 % of the lanes idle.
struct SlotTable {
  const void* nodes;          // the dense node -> slot table this was built from
  const uint32_t* slot_gi;    // dense index of the node that owns slot s; INVALID_NODE: nobody does
  const uint32_t* slot_yz;    // its y | z << 16
  uint32_t n_slots;           // highest used slot + 1
};

struct RowClasses {
  const void* map;             // the device node map the tables were built from
  const uint32_t* seg_class;   // bytes [arr_nz * arr_ny][nseg], addressed as dwords (scalar loads)
  const uint8_t* row_class;    // [arr_nz * arr_ny]
  const uint32_t* bc_rows;     // y | z << 16 of the class-2 rows
  int nseg;                    // segments per row = ceil(nx / 64)
  int n_rows, n_bc_rows;       // real rows, class-2 rows
  long long n_fluid_segments, n_segments;
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
                      const int* dirs, int nd, unsigned long long base, long long col_stride, int ncols, long long row_stride,
                      int nrows, long long buf_k_stride, long long buf_row_stride, bool deliver_all, hipStream_t s);

hipError_t launch_macro(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                        const SweepArgs& a, hipStream_t s);

// Builds the tables of RowClasses for `map` (device buffers seg_class, row_class, bc_rows and the counters
// {class-2 rows, class-0 segments} are the caller's); the counts are read back by the caller.
hipError_t launch_build_slot_table(const Geometry& g, const void* nodes, uint32_t* slot_gi, uint32_t* slot_yz,
                                   uint32_t* max_slot, hipStream_t s);
hipError_t launch_classify_rows(const Geometry& g, const void* map, uint32_t* seg_class, uint8_t* row_class,
                                uint32_t* bc_rows, uint32_t* counters, int nseg, hipStream_t s);

// ---- binary Shan-Chen (slf_sc.hip) ----
// macro pass: a.dist_in = lattice 0, a.dist_out = lattice 1 (both read), writes rho, phi, v
hipError_t launch_sc_macro(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                           const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, hipStream_t s);
// collide-and-stream of lattice `grid_idx`: reads rho, phi, v
hipError_t launch_sc_sweep(const KernelSelector& sel, int grid_idx, Prop prop, const Geometry& g, const Physics& ph,
                           const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x,
                           hipStream_t s);
// both lattices in one pass: a.dist_in / dist_out = lattice 0, a.dist_in2 / dist_out2 = lattice 1
hipError_t launch_sc_fused(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph, const ShanChen& sc,
                           const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x, hipStream_t s);
// single-component Shan-Chen: density pass (a.dist_in -> a.rho) and the sweep with the force
hipError_t launch_scs_macro(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                            const ShanChen& sc, const SweepArgs& a, hipStream_t s);
hipError_t launch_scs_sweep(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph,
                            const ShanChen& sc, const SweepArgs& a, int y0, int y1, int z0, int z1, int block_x,
                            hipStream_t s);
hipError_t launch_sc_init(const KernelSelector& sel, const Geometry& g, const Physics& ph, void* dist1, void* dist2,
                          const void* rho, const void* phi, const void* const v[3], const void* nodes, hipStream_t s);

// ---- several steps inside one launch (slf_resident.hip): 2-D lattices ----
// src / dst: the raw distribution arrays ([0] = the array of the in-place pattern / copy A of the two-copy pattern, [1] =
// copy B) and where the tiles go after `steps` steps from iteration it0; tiles of tile_x x tile_y nodes, halo of `halo`
hipError_t launch_resident(const KernelSelector& sel, bool aa, const Geometry& g, const Physics& ph, const SweepArgs& a,
                           const void* const src[2], void* const dst[2], int it0, int steps, int tile_x, int tile_y, int halo,
                           hipStream_t s);
int resident_halo(bool aa, int it0, int steps);
size_t resident_lds_bytes(int q, int precision, bool aa, int win_x, int win_y);

}  // namespace slf
