// Aligned push of one (y, z) row, shared by the whole-row sweep kernels (slf_row.hip, slf_sc.hip).
//
// Thread t of the workgroup owns node x = t + 1 of the row; f[] holds the post-collision populations of
// that node.  The value f[I] of node x belongs to node x + e_x (same slot, push streaming): it is handed to
// the thread that owns the target x -- __shfl_up/down inside a wave64, one LDS word per direction between
// neighbouring waves, the periodic wrap as the cyclic continuation -- and stored there, so every store
// instruction of a wave covers whole 128-byte lines (the arrays are allocated with x = 1 on a line
// boundary).  A value is stored iff its *source* node is active (takes part in the sweep), whatever the
// target is; without in-sweep wrap along x the ghost columns are written by the two edge lanes.
// Must be called by every thread of the workgroup (it contains a barrier).
#pragma once
#include "slf_sweep.h"

namespace slf {

template <class L>
constexpr int count_x_dirs() {
  int n = 0;
  for (int i = 0; i < L::Q; i++) n += (L::ex(i) > 0) ? 1 : 0;
  return n;
}

template <class R>
__device__ __forceinline__ R shfl_up1(R v) { return __shfl_up(v, 1); }
template <class R>
__device__ __forceinline__ R shfl_down1(R v) { return __shfl_down(v, 1); }

template <class L, class R, bool GENERAL, int NT>
__device__ __forceinline__ void row_push(const Geometry& g, const R (&f)[L::Q], R* dout, size_t ds, uint32_t gi,
                                         int x, int nx, bool live, bool active, const AxisOff& oy,
                                         const AxisOff& oz) {
  constexpr int NW = 16;
  constexpr int NXD = count_x_dirs<L>();
  __shared__ R s_out_p[NW][NXD], s_out_m[NW][NXD], s_wrap_p[NXD], s_wrap_m[NXD];
  __shared__ int s_act_p[NW], s_act_m[NW], s_actw_p, s_actw_m;
  const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6;
  const bool wrapx = g.wrap[0] != 0;
  const AxisOff ox0 = {0, 0};
  // push: the value of node x travels to x + e_x and is stored by the thread that owns the target x
  {
    int kp = 0, km = 0;
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) > 0) {
        if (lane == 63) s_out_p[w][kp] = f[I];
        if (x == nx) s_wrap_p[kp] = f[I];
        kp++;
      }
      if constexpr (L::ex(I) < 0) {
        if (lane == 0) s_out_m[w][km] = f[I];
        if (x == 1) s_wrap_m[km] = f[I];
        km++;
      }
    });
    if constexpr (GENERAL) {
      if (lane == 63) s_act_p[w] = (int)active;
      if (lane == 0) s_act_m[w] = (int)active;
      if (x == nx) s_actw_p = (int)active;
      if (x == 1) s_actw_m = (int)active;
    }
  }
  __syncthreads();
  bool from_left = live, from_right = live;   // is the node at x - 1 / x + 1 a source?
  if constexpr (GENERAL) {
    int a = __shfl_up((int)active, 1);
    if (lane == 0 && w > 0) a = s_act_p[w - 1];
    if (x == 1) a = wrapx ? s_actw_p : 0;
    from_left = a != 0;
    a = __shfl_down((int)active, 1);
    if (lane == 63) a = s_act_m[(w + 1) & (NW - 1)];
    if (x == nx) a = wrapx ? s_actw_m : 0;
    from_right = a != 0;
  } else {
    if (!wrapx) {
      if (x == 1) from_left = false;
      if (x == nx) from_right = false;
    }
  }
  {
    int kp = 0, km = 0;
    static_for<0, L::Q>([&](auto I) {
      const int off = dir_offset<L, I>(ox0, oy, oz, true);
      R* dst = dout + ds * (size_t)I + (uint32_t)((int)gi + off);
      R t = f[I];
      bool src_ok = active;
      if constexpr (L::ex(I) > 0) {
        // edge lane, no wrap: the value leaves the row into the ghost column x = nx + 1
        if (!wrapx && x == nx && active) st<0>(dst + 1, f[I]);
        t = shfl_up1<R>(f[I]);
        if (lane == 0 && w > 0) t = s_out_p[w - 1][kp];
        if (x == 1) t = s_wrap_p[kp];
        src_ok = from_left;
        kp++;
      }
      if constexpr (L::ex(I) < 0) {
        if (!wrapx && x == 1 && active) st<0>(dst - 1, f[I]);
        t = shfl_down1<R>(f[I]);
        if (lane == 63) t = s_out_m[(w + 1) & (NW - 1)][km];
        if (x == nx) t = s_wrap_m[km];
        src_ok = from_right;
        km++;
      }
      if (live && src_ok) st<NT>(dst, t);
    });
  }
}

}  // namespace slf
