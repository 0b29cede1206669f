// Aligned push of one (y, z) row, shared by the whole-row sweep kernels (slf_row.hip, slf_sc.hip).
//
// Thread t of the workgroup owns node x = t + 1 of the row; f[] holds the post-collision populations of
// that node.  The value f[I] of node x belongs to node x + e_x (same slot, push streaming): it is handed to
// the thread that owns the target x -- __shfl_up/down inside a wave64, one LDS word per direction between
// neighbouring waves, the periodic wrap as the cyclic continuation -- and stored there, so every store
// instruction of a wave covers whole 128-byte lines (the arrays are allocated with x = 1 on a line
// boundary).  A value is stored iff its *source* node is active (takes part in the sweep), whatever the
// target is; without in-sweep wrap along x the ghost columns are written by the two edge lanes.
// Must be called by every thread of the workgroup (it contains a barrier); thread t of workgroup b owns
// x = 1 + b * blockDim.x + t.
#pragma once
#include "slf_sweep.h"

namespace slf {

template <class R>
__device__ __forceinline__ R shfl_up1(R v) { return __shfl_up(v, 1); }
template <class R>
__device__ __forceinline__ R shfl_down1(R v) { return __shfl_down(v, 1); }

template <class L, class R, bool GENERAL, int NT>
__device__ __forceinline__ void row_push(const Geometry& g, const R (&f)[L::Q], R* dout, size_t ds, uint32_t row,
                                         uint32_t xi, int x, int nx, bool live, bool active, const AxisOff& oy,
                                         const AxisOff& oz, R* const* xsend = nullptr, const FaceRows* fr = nullptr) {
  constexpr int NW = 16;
  constexpr int NXD = count_x_dirs<L>();
  __shared__ R s_out_p[NW][NXD], s_out_m[NW][NXD], s_wrap_p[NXD], s_wrap_m[NXD];
  __shared__ int s_act_p[NW], s_act_m[NW], s_actw_p, s_actw_m;
  const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6;
  const bool wrapx = g.wrap[0] != 0;
  const AxisOff ox0 = {0, 0};
  // x-segment [xs, xe] of the row this workgroup owns.  One segment (the usual case): the wrap is part of the
  // in-workgroup exchange.  Several segments (rows whose wave count does not tile a CU, nx > 1024): what
  // leaves a segment is stored by its own edge lane, exactly like the ghost columns of a non-periodic row.
  const int xs = 1 + (int)(blockIdx.x * blockDim.x);
  const int xe = (xs + (int)blockDim.x - 1 < nx) ? xs + (int)blockDim.x - 1 : nx;
  const bool multi = gridDim.x > 1;
  const bool edge_stores = multi || !wrapx;
  // push: the value of node x travels to x + e_x and is stored by the thread that owns the target x
  {
    int kp = 0, km = 0;
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) > 0) {
        if (lane == 63) s_out_p[w][kp] = f[I];
        if (x == xe) s_wrap_p[kp] = f[I];
        kp++;
      }
      if constexpr (L::ex(I) < 0) {
        if (lane == 0) s_out_m[w][km] = f[I];
        if (x == xs) s_wrap_m[km] = f[I];
        km++;
      }
    });
    if constexpr (GENERAL) {
      if (lane == 63) s_act_p[w] = (int)active;
      if (lane == 0) s_act_m[w] = (int)active;
      if (x == xe) s_actw_p = (int)active;
      if (x == xs) s_actw_m = (int)active;
    }
  }
  __syncthreads();
  bool from_left = live, from_right = live;   // is the node at x - 1 / x + 1 a source?
  if constexpr (GENERAL) {
    int a = __shfl_up((int)active, 1);
    if (lane == 0 && w > 0) a = s_act_p[w - 1];
    if (x == xs) a = edge_stores ? 0 : s_actw_p;
    from_left = a != 0;
    a = __shfl_down((int)active, 1);
    if (lane == 63) a = s_act_m[(w + 1) & (NW - 1)];
    if (x == xe) a = edge_stores ? 0 : s_actw_m;
    from_right = a != 0;
  } else {
    if (edge_stores) {
      if (x == xs) from_left = false;
      if (x == xe) from_right = false;
    }
  }
  {
    int kp = 0, km = 0;
    static_for<0, L::Q>([&](auto I) {
      const int off = dir_offset<L, I>(ox0, oy, oz, true);     // wave-uniform: base in SGPRs, lane offset = x
      SLF_GLOBAL R* dst = at_byte(uniform_base(dout + ds * (size_t)I + (uint32_t)((int)row + off)),
                                  xi * (uint32_t)sizeof(R));
      R t = f[I];
      bool src_ok = active;
      if constexpr (L::ex(I) > 0) {
        // edge lane: the value leaves the segment -- into the next segment, the ghost column x = nx + 1, or
        // around the periodic seam to x = 1
        if (edge_stores && x == xe && active) {
          if (xsend && xsend[1] && x == nx) {      // leaves through a connected high face: straight into the send buffer
            xsend[1][face_elem<L, I>(*fr, 1)] = f[I];
          } else if (x == nx && !wrapx && (g.x_ghost_unused & 2)) {
            // a ghost column nothing reads
          } else {
            stg<0>(dst + ((wrapx && x == nx) ? -(nx - 1) : 1), f[I]);
          }
        }
        t = shfl_up1<R>(f[I]);
        if (lane == 0 && w > 0) t = s_out_p[w - 1][kp];
        if (x == xs) t = s_wrap_p[kp];
        src_ok = from_left;
        kp++;
      }
      if constexpr (L::ex(I) < 0) {
        if (edge_stores && x == xs && active) {
          if (xsend && xsend[0] && x == 1) {
            xsend[0][face_elem<L, I>(*fr, 1)] = f[I];
          } else if (x == 1 && !wrapx && (g.x_ghost_unused & 1)) {
            // a ghost column nothing reads
          } else {
            stg<0>(dst + ((wrapx && x == 1) ? (nx - 1) : -1), f[I]);
          }
        }
        t = shfl_down1<R>(f[I]);
        if (lane == 63) t = s_out_m[(w + 1) & (NW - 1)][km];
        if (x == xe) t = s_wrap_m[km];
        src_ok = from_right;
        km++;
      }
      if (live && src_ok) stg<NT>(dst, t);
    });
  }
}

// Workgroup width for a row of nx nodes.  Up to 1024 nodes: the whole row (rows of 9-12 waves run 15-25 % below the
// 8-wave rows of nx = 512, but cutting them into x-segments of 8 waves is slower still -- the partial-line stores of
// the segment edges cost more than the better fit buys: profiles/r01/segmented_rows.log vs pad_rowshape.log).  Longer
// rows have no whole-row form: equal segments of at most 8 waves, each a multiple of 64 nodes so that segment starts
// stay line aligned -- 5.9-6.1 TB/s on the x-streaming steps where the per-node kernel with its misaligned stores
// reaches 5.2-5.5 (profiles/r02/long_rows.log).
static inline int row_block_x(int nx) {
  const int waves = (nx + 63) / 64;
  if (nx <= 1024) return waves * 64;
  const int nseg = (waves + 7) / 8;
  return ((waves + nseg - 1) / nseg) * 64;
}

}  // namespace slf
