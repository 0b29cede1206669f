// Several time steps inside one launch for subdomains that are launch-bound ("CollideAndPropagateResident").
//
// A 256 x 256 D2Q9 subdomain (BASELINE config 1: examples/ldc_2d.py) is 65 k nodes: one sweep moves 4.7 MB and lasts
// as long as a launch does (4.7 us per step from a 16-step HIP graph, 0.12 of the HBM roofline).  The reference steps
// such a case with one launch per step as well (subdomain_runner.py:960-974); here a workgroup keeps a WINDOW of the
// subdomain -- its tile plus a halo of H nodes on every side -- in LDS and performs T steps on it before anything
// goes back to memory: one launch per T steps.
//
// Exact by construction: the window is a cache of the RAW slots of the distribution arrays (every slot of every node
// of the window, ghost nodes and never-written slots included), and a step on it is the step the per-node kernel
// (slf_kernels.hip: sweep_kernel) performs on memory -- the same node_update(), the same slots read and written, the
// same nodes excluded: in-place steps read the node's own slots and write their opposites (even iteration) or pull
// from / push to the neighbours (odd iteration), two-copy steps read copy (it & 1) and push into the other one.  What
// lies beyond the window's edge is missing, so a step is performed only by the nodes whose result the tile still
// depends on -- a rectangle around the tile that shrinks by two nodes per odd in-place step and by one per two-copy step
// (the kernel's comment on `lim`; the threads are dealt out over that rectangle, so the waves that work are full) -- and
// the host chooses the halo H so that the first step's rectangle fits
// (resident_halo()); after the T steps only the TILE -- exact -- is written back, every slot of every node of it.  Tiles cover the whole lattice box (ghost layer included; real nodes only along an
// axis wrapped in-sweep, where the window wraps too), each node belongs to exactly one tile.
//
// Source and destination are different buffers (a workgroup's halo is another workgroup's tile): the host
// alternates between the arrays and a scratch copy.  2-D lattices (a 3-D window of any depth does not fit into LDS).
// Not served (refused by slf_kernel_set_args / the host falls back to stepping): half-way bounce-back and outflow
// nodes (their node code writes / reads memory directly), indirect addressing, --minimize_roundoff, Shan-Chen models.
#include "../../include/sailfish_hip.h"
#include "slf_kernels.h"
#include "slf_node.h"
#include "slf_sweep.h"

namespace slf {


template <class L, class R>
struct ResidentParams {
  SweepParams<L, R> p;          // map, node_params, status, options, g, cp (din / dout unused)
  const R* src[2];              // raw distribution arrays (two-copy pattern: copy A, copy B; in place: [0] only)
  R* dst[2];                    // where the tiles go
  int it0, steps;               // first iteration, number of steps
  int tile[2], halo, win[2];    // tile, halo and window extents (win = tile + 2 halo)
  int org[2];                   // lattice coordinate of the first tile's first node per axis (1 on a wrapped axis)
  int ext[2];                   // nodes to cover per axis (lat - 2 on a wrapped axis, lat otherwise)
};

// window node -> lattice coordinate along one axis; false: beyond the box
__device__ __forceinline__ bool resident_coord(int c, int lat, int wrap, int& out) {
  if (wrap) {
    const int n = lat - 2;
    int r = (c - 1) % n;
    if (r < 0) r += n;
    out = r + 1;
    return true;
  }
  out = c;
  return c >= 0 && c < lat;
}

// per window node, decided once when the window is loaded
constexpr uint32_t RI_ACTIVE = 1u;     // a node the sweep works on (real, not excluded by its node code)
constexpr uint32_t RI_SIMPLE = 2u;     // ... whose code is the plain one: fluid or full-way bounce-back (node_update<BCL = 0>)
constexpr uint32_t RI_TILE = 4u;       // node of the tile (the on-GPU invalid-value check looks at these only)
constexpr uint32_t RI_WALL = 8u;       // a plain node that is a full-way bounce-back node
constexpr int RI_DIST_SHIFT = 8;       // bits 8..: distance from the tile (0 inside it)
// (s_complex + s_ncomplex below: 4100 bytes of STATIC LDS next to the window -- the hosts' 160 KiB bound counts 4352 for it)
constexpr int RESIDENT_MAX_COMPLEX = 2048;  // boundary-condition nodes of a window (all of them fit: windows have at most 2048 nodes)

template <class L, class R, int MODEL, bool AA, bool GENERAL>
__global__ void __launch_bounds__(1024) resident_kernel(const ResidentParams<L, R> rp) {
  static_assert(L::dim == 2, "2-D lattices");
  extern __shared__ unsigned char resident_lds[];
  const SweepParams<L, R>& p = rp.p;
  const Geometry& g = p.g;
  const int WW = rp.win[0], WH = rp.win[1];
  const int NW = WW * WH;
  constexpr int Q = L::Q;
  // LDS: two [q][window node] blocks of populations, then node codes and node flags.  Two-copy pattern: the blocks are
  // copy A and copy B.  In place: block 0 is the array; block 1 takes the post-collision populations of an odd step
  // until every pull of the step is done (the push must not overtake the pull of a neighbour).
  R* const win = (R*)resident_lds;
  R* const win2 = win + (size_t)Q * NW;
  uint32_t* const codes = (uint32_t*)(win2 + (size_t)Q * NW);
  uint32_t* const info = codes + NW;
  // boundary-condition nodes: their long node code runs in ONE wave, over a list, instead of in every wave that happens
  // to hold one of them (a wave pays for the code of its slowest lane)
  __shared__ uint32_t s_ncomplex;
  __shared__ uint16_t s_complex[RESIDENT_MAX_COMPLEX];
  if (threadIdx.x == 0) s_ncomplex = 0;
  __syncthreads();
  const int tx0 = rp.org[0] + (int)blockIdx.x * rp.tile[0];
  const int ty0 = rp.org[1] + (int)blockIdx.y * rp.tile[1];
  const size_t ds = g.dist_size;
  const int nthreads = (int)blockDim.x;

  // ---- load the window: every slot of every node, as it lies in memory
  for (int n = (int)threadIdx.x; n < NW; n += nthreads) {
    const int wx = n % WW, wy = n / WW;
    int gx, gy;
    const bool in = resident_coord(tx0 - rp.halo + wx, g.lat_nx, g.wrap[0], gx) &
                    resident_coord(ty0 - rp.halo + wy, g.lat_ny, g.wrap[1], gy);
    uint32_t code = 0, flags = 0;
    {
      // distance from the tile (Chebyshev)
      const int dx = wx < rp.halo ? rp.halo - wx : (wx >= rp.halo + rp.tile[0] ? wx - (rp.halo + rp.tile[0] - 1) : 0);
      const int dy = wy < rp.halo ? rp.halo - wy : (wy >= rp.halo + rp.tile[1] ? wy - (rp.halo + rp.tile[1] - 1) : 0);
      flags = (uint32_t)(dx > dy ? dx : dy) << RI_DIST_SHIFT;
    }
    if (in) {
      const uint32_t gi = (uint32_t)gx + (uint32_t)g.arr_nx * (uint32_t)gy;
      // the sweep covers the real nodes 1 .. lat - 2; what their node code says decides the rest
      bool active = gx >= 1 && gx <= g.lat_nx - 2 && gy >= 1 && gy <= g.lat_ny - 2;
      bool simple = true;
      if constexpr (GENERAL) {
        if (active) {
          code = p.map[gi];
          const int kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
          active = !kind_is_excluded(kind);
          simple = kind == NK_FLUID || kind == NK_FULL_BB;
        }
      }
      if (active) {
        flags |= RI_ACTIVE;
        if (simple) {
          flags |= RI_SIMPLE;
          if constexpr (GENERAL) {
            if ((int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull) == NK_FULL_BB) flags |= RI_WALL;
          }
        } else {
          s_complex[atomicAdd(&s_ncomplex, 1u)] = (uint16_t)n;
        }
        if (wx >= rp.halo && wx < rp.halo + rp.tile[0] && wy >= rp.halo && wy < rp.halo + rp.tile[1]) flags |= RI_TILE;
      }
      static_for<0, Q>([&](auto I) { win[(size_t)I * NW + n] = (rp.src[0] + ds * (size_t)I)[gi]; });
      if constexpr (!AA) {
        static_for<0, Q>([&](auto I) { win2[(size_t)I * NW + n] = (rp.src[1] + ds * (size_t)I)[gi]; });
      }
    } else {
      static_for<0, Q>([&](auto I) { win[(size_t)I * NW + n] = (R)0; });
      if constexpr (!AA) {
        static_for<0, Q>([&](auto I) { win2[(size_t)I * NW + n] = (R)0; });
      }
    }
    codes[n] = code;
    info[n] = flags;
  }
  __syncthreads();

  // window offsets of the neighbours (no wrap inside the window: it wraps where it is loaded)
  auto woff = [&](auto I) -> int { return L::ex(I) + L::ey(I) * WW; };
  const AxisOff none = {0, 0};

  const int TW = rp.tile[0], TH = rp.tile[1];
  const int ncomplex = (int)s_ncomplex;

  // one node: read, node code, write -- all three for a compile-time MODE: 0 = local (even in-place iteration: own slots in,
  // opposite own slots out), 1 = pull, post-collision populations staged in block 1 (odd in-place iteration), 2 / 3 =
  // two-copy push from block 0 to 1 / from 1 to 0.  BCL = 0: the plain node code (fluid, full-way bounce-back; the kind
  // comes from the node's flags), BCL = 2: everything.  One instantiation of node_update() per BCL for every step kind:
  // the propagation mode only matters to node kinds this kernel does not serve.
  auto do_node = [&](auto MODE, auto BCL, int n, uint32_t flags) {
    constexpr int mode = decltype(MODE)::value;
    const R* const rd = (mode == 3) ? win2 : win;
    R* const wr = (mode == 0 || mode == 3) ? win : win2;
    R f[Q];
    if constexpr (mode == 1) {
      static_for<0, Q>([&](auto I) { f[I] = rd[(size_t)L::opp(I) * NW + (n - woff(I))]; });
    } else {
      static_for<0, Q>([&](auto I) { f[I] = rd[(size_t)I * NW + n]; });
    }
    uint32_t code = 0u;
    int kind = (flags & RI_WALL) ? NK_FULL_BB : NK_FLUID;
    if constexpr (GENERAL && decltype(BCL)::value != 0) {
      code = codes[n];
      kind = (int)((g.type_lut >> (4u * (code & g.type_mask))) & 0xFull);
    }
    R rho, v[3];
    bool wet = true;
    node_update<L, R, MODEL, PROP_AA_ODD, GENERAL, false, FORCE_RUNTIME, decltype(BCL)::value>(p, f, code, kind, 0u, none, none, none,
                                                                                           rho, v, wet);
    if (wet && (p.options & OPTION_CHECK_INVALID) && (flags & RI_TILE)) {
      int gx, gy;
      resident_coord(tx0 - rp.halo + n % WW, g.lat_nx, g.wrap[0], gx);
      resident_coord(ty0 - rp.halo + n / WW, g.lat_ny, g.wrap[1], gy);
      check_invalid<R>(p.status, p.options, rho, gx, gy, 0);
    }
    if constexpr (mode == 0) {
      static_for<0, Q>([&](auto I) { wr[(size_t)L::opp(I) * NW + n] = f[I]; });
    } else if constexpr (mode == 1) {
      static_for<0, Q>([&](auto I) { wr[(size_t)I * NW + n] = f[I]; });                  // staged at the node itself
    } else {
      static_for<0, Q>([&](auto I) { wr[(size_t)I * NW + (n + woff(I))] = f[I]; });
    }
  };

  // One step for a compile-time MODE.  The nodes that still matter are those up to `lim` from the tile -- a rectangle of
  // (TW + 2 lim) x (TH + 2 lim) nodes; thread t takes its t-th node (row by row), so that the waves that work are full
  // and the others skip the step: t / width through a multiplication with magic = ceil(2^32 / width) (exact for t < 2^16).
  auto do_step = [&](auto MODE, int lim) {
    constexpr int mode = decltype(MODE)::value;
    const int aw = TW + 2 * lim, ah = TH + 2 * lim;
    const int count = aw * ah;
    const uint32_t magic = 0xffffffffu / (uint32_t)aw + 1u;
    const int base = (rp.halo - lim) * WW + (rp.halo - lim);
#pragma unroll 1
    for (int t = (int)threadIdx.x; t < count; t += nthreads) {
      const int row = (int)__umulhi((uint32_t)t, magic);
      const int n = base + row * WW + (t - row * aw);
      const uint32_t flags = info[n];
      if ((flags & (RI_ACTIVE | RI_SIMPLE)) == (RI_ACTIVE | RI_SIMPLE)) do_node(MODE, std::integral_constant<int, 0>{}, n, flags);
    }
    // the listed boundary-condition nodes: ONE wave, the last of the workgroup (the first to run out of nodes above)
    if ((int)threadIdx.x >= nthreads - 64) {
#pragma unroll 1
      for (int c = (int)threadIdx.x - (nthreads - 64); c < ncomplex; c += 64) {
        const int n = (int)s_complex[c];
        const uint32_t flags = info[n];
        if ((int)(flags >> RI_DIST_SHIFT & 0xffu) > lim) continue;
        do_node(MODE, std::integral_constant<int, 2>{}, n, flags);
      }
    }
    __syncthreads();
    if constexpr (mode == 1) {
      // every pull is done: the staged populations go to the neighbours' slots
#pragma unroll 1
      for (int t = (int)threadIdx.x; t < count; t += nthreads) {
        const int row = (int)__umulhi((uint32_t)t, magic);
        const int n = base + row * WW + (t - row * aw);
        if (!(info[n] & RI_ACTIVE)) continue;
        static_for<0, Q>([&](auto I) { win[(size_t)I * NW + (n + woff(I))] = win2[(size_t)I * NW + n]; });
      }
      __syncthreads();
    }
  };

  int odd_after = 0;       // odd iterations among the steps after the current one (counted down)
  for (int q = 1; q < rp.steps; q++) odd_after += (rp.it0 + q) & 1;
  for (int s = 0; s < rp.steps; s++) {
    const int it = rp.it0 + s;
    const bool odd = (it & 1) != 0;
    // Which nodes still matter.  After the last step the tile (distance 0) must be exact.  For the nodes up to distance r
    // to be exact after a step that pushes, the nodes up to r + 1 perform it, from inputs exact up to r + 2 (in place:
    // they pull first) resp. r + 1 (two-copy); a local step needs the same region before as after.  So with r = what
    // the steps AFTER this one need, this step is performed by the nodes up to `lim` from the tile; the window's
    // outermost ring (distance = halo) is never among those of a pushing step.
    const int lim = AA ? 2 * odd_after + (odd ? 1 : 0) : rp.steps - s;
    if (s + 1 < rp.steps) odd_after -= (it + 1) & 1;
    if constexpr (AA) {
      if (odd) do_step(std::integral_constant<int, 1>{}, lim);
      else do_step(std::integral_constant<int, 0>{}, lim);
    } else {
      if (odd) do_step(std::integral_constant<int, 3>{}, lim);
      else do_step(std::integral_constant<int, 2>{}, lim);
    }
  }

  // ---- write the tile back: every slot of every node of it (nodes beyond the box belong to nobody)
  for (int t = (int)threadIdx.x; t < TW * TH; t += nthreads) {
    const int ix = t % TW, iy = t / TW;
    if (tx0 - rp.org[0] + ix >= rp.ext[0] || ty0 - rp.org[1] + iy >= rp.ext[1]) continue;   // the last tile of an axis may stick out
    const int n = (rp.halo + iy) * WW + rp.halo + ix;
    int gx, gy;
    resident_coord(tx0 + ix, g.lat_nx, g.wrap[0], gx);
    resident_coord(ty0 + iy, g.lat_ny, g.wrap[1], gy);
    const uint32_t gi = (uint32_t)gx + (uint32_t)g.arr_nx * (uint32_t)gy;
    static_for<0, Q>([&](auto I) { (rp.dst[0] + ds * (size_t)I)[gi] = win[(size_t)I * NW + n]; });
    if constexpr (!AA) {
      static_for<0, Q>([&](auto I) { (rp.dst[1] + ds * (size_t)I)[gi] = win2[(size_t)I * NW + n]; });
    }
  }
}

// How far what is wrong has spread from the window's ring after `steps` steps from iteration it0 (see the header):
// the halo the tile needs.
int resident_halo(bool aa, int it0, int steps) {
  if (!aa) return steps + 1;
  int odd = 0;
  for (int s = 0; s < steps; s++) odd += (it0 + s) & 1;
  return odd ? 2 * odd : 1;
}

size_t resident_lds_bytes(int q, int precision, bool aa, int win_x, int win_y) {
  (void)aa;      // in place: the array and the staging block of the odd steps; two-copy: copy A and copy B
  const size_t nw = (size_t)win_x * (size_t)win_y;
  return nw * (size_t)q * (size_t)precision * 2 + nw * 8;
}

template <class L, class R, int MODEL, bool AA>
static hipError_t launch_resident3(bool general, const ResidentParams<L, R>& rp, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  if (lds > 64 * 1024) {
    // beyond the default 64 KiB of dynamic LDS a kernel has to ask for it (160 KiB per workgroup on gfx950)
    const void* fn = general ? (const void*)resident_kernel<L, R, MODEL, AA, true> : (const void*)resident_kernel<L, R, MODEL, AA, false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (general) hipLaunchKernelGGL((resident_kernel<L, R, MODEL, AA, true>), grid, block, lds, s, rp);
  else hipLaunchKernelGGL((resident_kernel<L, R, MODEL, AA, false>), grid, block, lds, s, rp);
  return hipGetLastError();
}

template <class L, class R>
static hipError_t launch_resident2(const KernelSelector& sel, bool aa, const Geometry& g, const Physics& ph, const SweepArgs& a,
                                   const void* const src[2], void* const dst[2], int it0, int steps, int tile_x, int tile_y,
                                   int halo, hipStream_t s) {
  ResidentParams<L, R> rp;
  rp.p = make_params<L, R>(g, ph, a, 0, 0);
  for (int c = 0; c < 2; c++) {
    rp.src[c] = (const R*)src[c];
    rp.dst[c] = (R*)dst[c];
  }
  rp.it0 = it0;
  rp.steps = steps;
  rp.tile[0] = tile_x;
  rp.tile[1] = tile_y;
  rp.halo = halo;
  rp.win[0] = tile_x + 2 * halo;
  rp.win[1] = tile_y + 2 * halo;
  const int lat[2] = {g.lat_nx, g.lat_ny};
  for (int d = 0; d < 2; d++) {
    rp.org[d] = g.wrap[d] ? 1 : 0;
    rp.ext[d] = g.wrap[d] ? lat[d] - 2 : lat[d];
  }
  const int nw = rp.win[0] * rp.win[1];
  int threads = ((nw + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  dim3 block(threads, 1, 1);
  dim3 grid((rp.ext[0] + tile_x - 1) / tile_x, (rp.ext[1] + tile_y - 1) / tile_y, 1);
  const size_t lds = resident_lds_bytes(L::Q, (int)sizeof(R), aa, rp.win[0], rp.win[1]);
  if (sel.model == 0) {
    if (aa) return launch_resident3<L, R, 0, true>(sel.general, rp, grid, block, lds, s);
    return launch_resident3<L, R, 0, false>(sel.general, rp, grid, block, lds, s);
  }
  if (aa) return launch_resident3<L, R, 1, true>(sel.general, rp, grid, block, lds, s);
  return launch_resident3<L, R, 1, false>(sel.general, rp, grid, block, lds, s);
}

hipError_t launch_resident(const KernelSelector& sel, bool aa, const Geometry& g, const Physics& ph, const SweepArgs& a,
                           const void* const src[2], void* const dst[2], int it0, int steps, int tile_x, int tile_y, int halo,
                           hipStream_t s) {
  if (sel.lattice != 0) return hipErrorInvalidValue;
  if (sel.precision == 4) return launch_resident2<D2Q9, float>(sel, aa, g, ph, a, src, dst, it0, steps, tile_x, tile_y, halo, s);
  return launch_resident2<D2Q9, double>(sel, aa, g, ph, a, src, dst, it0, steps, tile_x, tile_y, halo, s);
}

}  // namespace slf
