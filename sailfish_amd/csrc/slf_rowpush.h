// Aligned push of one (y, z) row, shared by the whole-row sweep kernels (slf_row.hip, slf_sc.hip).
//
// Thread t of the workgroup owns node x = t + 1 of the row; f[] holds the post-collision populations of
// that node.  The value f[I] of node x belongs to node x + e_x (same slot, push streaming): it is handed to
// the thread that owns the target x -- a DPP wave shift inside a wave64, one LDS word per direction between
// neighbouring waves, the periodic wrap as the cyclic continuation -- and stored there, so every store
// instruction of a wave covers whole 128-byte lines (the arrays are allocated with x = 1 on a line
// boundary).  A value is stored iff its *source* node is active (takes part in the sweep), whatever the
// target is; without in-sweep wrap along x the ghost columns are written by the two edge lanes.
// Must be called by every thread of the workgroup (it contains a barrier); thread t of workgroup b owns
// x = 1 + b * blockDim.x + t.
#pragma once
#include "slf_sweep.h"

namespace slf {

// Wave-wide shift by one lane as ONE VALU instruction (v_mov_b32_dpp wave_shr:1 / wave_shl:1, gfx9): lane l receives
// the value of lane l - 1 (up) or l + 1 (down); the lane without a source -- lane 0 / lane 63 -- receives `edge`, the
// value that comes from the neighbouring wave.  (__shfl_up/down go through the LDS crossbar, ds_bpermute_b32, and the
// edge lane then costs a predicated LDS read: 15 instructions per direction where this needs 2.)
__device__ __forceinline__ int dpp_up1(int edge, int v) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_down1(int edge, int v) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false); }
template <class R, bool UP>
__device__ __forceinline__ R lane_shift1(R edge, R v) {
  if constexpr (sizeof(R) == 4) {
    const int e = __builtin_bit_cast(int, edge), x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(R, UP ? dpp_up1(e, x) : dpp_down1(e, x));
  } else {
    const long long e = __builtin_bit_cast(long long, edge), x = __builtin_bit_cast(long long, v);
    const int lo = UP ? dpp_up1((int)e, (int)x) : dpp_down1((int)e, (int)x);
    const int hi = UP ? dpp_up1((int)(e >> 32), (int)(x >> 32)) : dpp_down1((int)(e >> 32), (int)(x >> 32));
    return __builtin_bit_cast(R, ((long long)hi << 32) | (long long)(unsigned)lo);
  }
}

template <class L, class R, bool GENERAL, int NT>
__device__ __forceinline__ void row_push(const Geometry& g, const R (&f)[L::Q], R* dout, size_t ds, uint32_t row,
                                         uint32_t xi, int x, int nx, bool live, bool active, const AxisOff& oy,
                                         const AxisOff& oz, R* const* xsend = nullptr, const FaceRows* fr = nullptr) {
  constexpr int NW = 16;
  constexpr int NXD = count_x_dirs<L>();
  // what leaves a wave through its last / first lane, per x-moving direction; slot NW: what leaves the segment (the
  // periodic wrap, read by the lane at the other end)
  __shared__ R s_out_p[NW + 1][NXD], s_out_m[NW + 1][NXD];
  __shared__ int s_act_p[NW + 1], s_act_m[NW + 1];
  const int lane = (int)threadIdx.x & 63;
  const int w = sgpr((int)threadIdx.x >> 6);
  const bool wrapx = g.wrap[0] != 0;
  const AxisOff ox0 = {0, 0};
  // x-segment [xs, xe] of the row this workgroup owns.  One segment (the usual case): the wrap is part of the
  // in-workgroup exchange.  Several segments (rows whose wave count does not tile a CU, nx > 1024): what
  // leaves a segment is stored by its own edge lane, exactly like the ghost columns of a non-periodic row.
  const int xs = 1 + (int)(blockIdx.x * blockDim.x);
  const int xe = (xs + (int)blockDim.x - 1 < nx) ? xs + (int)blockDim.x - 1 : nx;
  const bool multi = gridDim.x > 1;
  const bool edge_stores = multi || !wrapx;
  const bool is_xs = x == xs, is_xe = x == xe;
  const int wlast = (xe - xs) >> 6;          // the wave that owns x = xe
  // ---- hand over: one predicated block per edge (not per direction: every block costs an exec-mask round trip)
  if (lane == 63) {
    int k = 0;
    static_for<1, L::Q>([&](auto I) { if constexpr (L::ex(I) > 0) s_out_p[w][k++] = f[I]; });
    if constexpr (GENERAL) s_act_p[w] = (int)active;
  }
  if (is_xe) {
    int k = 0;
    static_for<1, L::Q>([&](auto I) { if constexpr (L::ex(I) > 0) s_out_p[NW][k++] = f[I]; });
    if constexpr (GENERAL) s_act_p[NW] = (int)active;
  }
  if (lane == 0) {
    int k = 0;
    static_for<1, L::Q>([&](auto I) { if constexpr (L::ex(I) < 0) s_out_m[w][k++] = f[I]; });
    if constexpr (GENERAL) s_act_m[w] = (int)active;
  }
  if (is_xs) {
    int k = 0;
    static_for<1, L::Q>([&](auto I) { if constexpr (L::ex(I) < 0) s_out_m[NW][k++] = f[I]; });
    if constexpr (GENERAL) s_act_m[NW] = (int)active;
  }
  __syncthreads();
  // where the edge lanes of this wave read: the neighbouring wave's slot, or the wrap slot at the segment's ends
  // (wave-uniform: every lane reads the same word, a broadcast)
  const int slot_p = sgpr(w == 0 ? NW : w - 1);
  const int slot_m = sgpr(w == wlast ? NW : w + 1);
  bool from_left = live, from_right = live;   // is the node at x - 1 / x + 1 a source?
  if constexpr (GENERAL) {
    from_left = dpp_up1(s_act_p[slot_p], (int)active) != 0;
    int a = dpp_down1(s_act_m[slot_m], (int)active);
    if (is_xe) a = s_act_m[NW];               // x = xe need not be lane 63 (nx % 64 != 0)
    from_right = a != 0;
  }
  if (edge_stores) {
    if (is_xs) from_left = false;
    if (is_xe) from_right = false;
  }
  // destination of population I in this thread's node: wave-uniform row base in SGPRs, lane offset = x
  auto dst_of = [&](auto I) -> SLF_GLOBAL R* {
    const int off = dir_offset<L, I>(ox0, oy, oz, true);
    return at_byte(uniform_base(dout + ds * (size_t)I + (uint32_t)((int)row + off)), xi * (uint32_t)sizeof(R));
  };
  // ---- edge lanes: the value leaves the segment -- into the next segment, the ghost column x = nx + 1 / 0, a
  // connected face's send buffer, or around the periodic seam of a multi-segment row
  if (edge_stores) {
    if (is_xe && active) {
      static_for<1, L::Q>([&](auto I) {
        if constexpr (L::ex(I) > 0) {
          if (xsend && xsend[1] && x == nx) {      // leaves through a connected high face: straight into the send buffer
            xsend[1][face_elem<L, I>(*fr, 1)] = f[I];
          } else if (x == nx && !wrapx && (g.x_ghost_unused & 2)) {
            // a ghost column nothing reads
          } else {
            stg<0>(dst_of(I) + ((wrapx && x == nx) ? -(nx - 1) : 1), f[I]);
          }
        }
      });
    }
    if (is_xs && active) {
      static_for<1, L::Q>([&](auto I) {
        if constexpr (L::ex(I) < 0) {
          if (xsend && xsend[0] && x == 1) {
            xsend[0][face_elem<L, I>(*fr, 1)] = f[I];
          } else if (x == 1 && !wrapx && (g.x_ghost_unused & 1)) {
            // a ghost column nothing reads
          } else {
            stg<0>(dst_of(I) + ((wrapx && x == 1) ? (nx - 1) : -1), f[I]);
          }
        }
      });
    }
  }
  // ---- the stores, one predicated block per class of directions.  A value is stored iff its source node is active.
  if (live && active) {
    static_for<0, L::Q>([&](auto I) { if constexpr (L::ex(I) == 0) stg<NT>(dst_of(I), f[I]); });
  }
  {
    R t[NXD];
    int k = 0;
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) > 0) { t[k] = lane_shift1<R, true>(s_out_p[slot_p][k], f[I]); k++; }
    });
    if (live && from_left) {
      k = 0;
      static_for<1, L::Q>([&](auto I) { if constexpr (L::ex(I) > 0) stg<NT>(dst_of(I), t[k++]); });
    }
  }
  {
    R t[NXD];
    int k = 0;
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) < 0) {
        const R edge = s_out_m[slot_m][k];
        t[k] = lane_shift1<R, false>(edge, f[I]);
        if (is_xe) t[k] = s_out_m[NW][k];     // x = xe need not be lane 63 (nx % 64 != 0)
        k++;
      }
    });
    if (live && from_right) {
      k = 0;
      static_for<1, L::Q>([&](auto I) { if constexpr (L::ex(I) < 0) stg<NT>(dst_of(I), t[k++]); });
    }
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
