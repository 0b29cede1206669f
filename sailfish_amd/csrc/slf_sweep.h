// Shared between slf_kernels.hip (general kernels) and slf_fast.hip (the tuned
// D3Q19 / f32 / fluid-only north-star kernels): parameter block and neighbour
// offset helpers of the sweep.
#pragma once
#include "slf_kernels.h"
#include "slf_node.h"

namespace slf {

template <class L, class R>
struct SweepParams {
  const uint32_t* __restrict__ map;
  const R* din;
  R* dout;
  R* rho;
  R* vx;
  R* vy;
  R* vz;
  const R* __restrict__ node_params;
  uint32_t options;
  int y0, z0;
  int relaxation_enabled;
  Geometry g;
  CollideParams<L, R> cp;
};

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

template <class L, class R>
inline SweepParams<L, R> make_params(const Geometry& g, const Physics& ph, const SweepArgs& a, int y0, int z0) {
  SweepParams<L, R> p;
  p.map = (const uint32_t*)a.map;
  p.din = (const R*)a.dist_in;
  p.dout = (R*)a.dist_out;
  p.rho = (R*)a.rho;
  p.vx = (R*)a.v[0];
  p.vy = (R*)a.v[1];
  p.vz = (R*)a.v[2];
  p.node_params = (const R*)a.node_params;
  p.options = a.options;
  p.y0 = y0;
  p.z0 = z0;
  p.relaxation_enabled = ph.relaxation_enabled;
  p.g = g;
  p.cp.omega = (R)(1.0 / ph.tau);
  for (int k = 0; k < L::Q; k++) p.cp.mrt_s[k] = (R)ph.mrt_rates[k];
  for (int d = 0; d < 3; d++) p.cp.accel[d] = (R)ph.accel[d];
  p.cp.guo_pref = (R)(3.0 * (1.0 - 0.5 / ph.tau));
  p.cp.incompressible = ph.incompressible;
  p.cp.has_force = ph.has_force;
  return p;
}


// Returns true when the launch was handled by a tuned kernel (status in *err).
bool launch_sweep_fast(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph, const SweepArgs& a,
                       int y0, int y1, int z0, int z1, int block_x, hipStream_t s, hipError_t* err);

}  // namespace slf
