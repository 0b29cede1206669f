// Tuned sweep kernels for the north-star configuration: D3Q19, single
// precision, fluid-only subdomain (no node map), x wrapped inside the sweep.
// Same arithmetic (slf_node.h) and same memory layout as the general kernels in
// slf_kernels.hip -- results are bit-identical; only the access shape differs.
//
// Geometry.variant bits:
//   1   these kernels at all (0 = the general per-node kernels of slf_kernels.hip)
//   2   even AA step processes 2 consecutive x nodes per thread with 8-byte accesses
//       (the even step touches only the node's own slots -> every access is aligned)
//   8   x-streaming steps (odd AA, AB) use the whole-row kernel with aligned accesses
// Populations are streamed exactly once per step: every access carries the non-temporal hint (NT).
#include "slf_rowpush.h"

namespace slf {

typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int NT = 3;   // non-temporal loads and stores

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef float2v type; };

template <int VEC>
__device__ __forceinline__ float vget(const typename VecT<VEC>::type& v, int k) {
  if constexpr (VEC == 1) return v;
  else return v[k];
}
template <int VEC>
__device__ __forceinline__ void vset(typename VecT<VEC>::type& v, int k, float x) {
  if constexpr (VEC == 1) v = x;
  else v[k] = x;
}

// Even AA step (own node, opposite slot), VEC nodes per thread.  Thread 0 of a row starts at the first
// real node x = 1; the launcher only selects VEC > 1 when x = 1 is VEC-element aligned in memory (the
// backend allocates distribution arrays with that offset), so every access is an aligned 8 / 16-byte one.
template <int MODEL, int VEC, bool FORCE, bool XFACE>
__device__ __forceinline__ void fast_even_body(const SweepParams<D3Q19, float>& p, int gy, int gz, uint32_t vi, int gx0) {
  using L = D3Q19;
  typedef typename VecT<VEC>::type V;
  const Geometry& g = p.g;
  const uint32_t row = sgpr((uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz);
  const uint32_t gi = row + (uint32_t)gx0;
  const uint32_t vb = vi * (uint32_t)sizeof(V);                // ... and its byte offset from x = 1
  const size_t ds = g.dist_size;
  const size_t drow = row;
  V fv[L::Q];
  static_for<0, L::Q>([&](auto I) {
    fv[I] = ldg<NT>(at_byte(uniform_base((const V*)(p.din + ds * (size_t)I + drow + 1)), vb));
  });
  V orho, ovx, ovy, ovz;
#pragma unroll
  for (int k = 0; k < VEC; k++) {
    float f[L::Q];
    static_for<0, L::Q>([&](auto I) { f[I] = vget<VEC>(fv[I], k); });
    if constexpr (XFACE) {     // x faces connected through face buffers (x not wrapped): edge nodes only
      const FaceRows fr = face_rows<L>(g, gy, gz);
      x_face_receive<L, float, false>(p, f, gx0 + k, g.lat_nx - 2, fr);
    }
    float rho, v[3];
    macro_standard<L, float>(f, p.cp.incompressible != 0, rho, v);
    if (gx0 + k <= g.lat_nx - 2) check_invalid<float>(p.status, p.options, rho, gx0 + k, gy, gz);
    if (p.relaxation_enabled) {
      if constexpr (MODEL == 0) bgk_relax<L, float, FORCE>(f, rho, v, p.cp);
      else mrt_relax<L, float, FORCE>(f, v, p.cp, false);
    }
    if constexpr (XFACE) {
      const FaceRows fr = face_rows<L>(g, gy, gz);
      x_face_send_own_row<L, float>(p, f, gx0 + k, g.lat_nx - 2, fr);
    }
    static_for<0, L::Q>([&](auto I) { vset<VEC>(fv[I], k, f[I]); });
    vset<VEC>(orho, k, rho);
    vset<VEC>(ovx, k, v[0]);
    vset<VEC>(ovy, k, v[1]);
    vset<VEC>(ovz, k, v[2]);
  }
  if (p.options & 1u) {
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      if (gx0 + k <= g.lat_nx - 2) {
        p.rho[gi + k] = vget<VEC>(orho, k);
        p.vx[gi + k] = vget<VEC>(ovx, k);
        p.vy[gi + k] = vget<VEC>(ovy, k);
        p.vz[gi + k] = vget<VEC>(ovz, k);
      }
    }
  }
  // nodes of the last group beyond x = nx are the high ghost / padding: their own slots are never read
  // when x is wrapped in-sweep (without the wrap the launcher asks for nx % VEC == 0)
  static_for<0, L::Q>([&](auto I) {
    stg<NT>(at_byte(uniform_base((V*)(p.dout + ds * (size_t)L::opp(I) + drow + 1)), vb), fv[I]);
  });
}

template <int MODEL, int VEC, bool FORCE, bool XFACE>
// (second launch bound: minimum waves per SIMD.  The force-free BGK instantiation needs 82 VGPRs when left alone,
// two more than six resident waves allow.)
__global__ void __launch_bounds__(512, (VEC == 2 && MODEL == 0 && !FORCE) ? 6 : 2) fast_even_kernel(const SweepParams<D3Q19, float> p) {
  const Geometry& g = p.g;
  const int gy = sgpr(p.y0 + (XFACE ? xcd_row((int)blockIdx.y, p.row_mode >> 4) : (int)blockIdx.y));
  const int gz = sgpr(p.z0 + (int)blockIdx.z);
  const uint32_t vi = blockIdx.x * blockDim.x + threadIdx.x;   // index of this thread's VEC-node group in the row
  const int gx0 = 1 + (int)vi * VEC;
  if (gx0 > g.lat_nx - 2) return;
  fast_even_body<MODEL, VEC, FORCE, XFACE>(p, gy, gz, vi, gx0);
}

// One node per thread, any propagation mode (scalar accesses), optional non-temporal hint.
template <int MODEL, int PROP, bool FORCE>
__global__ void __launch_bounds__(1024) fast_scalar_kernel(const SweepParams<D3Q19, float> p) {
  using L = D3Q19;
  const Geometry& g = p.g;
  const int gy = p.y0 + (int)blockIdx.y;
  const int gz = p.z0 + (int)blockIdx.z;
  const int gx = 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (gx > g.lat_nx - 2) return;
  const uint32_t gi = (uint32_t)gx + (uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz;
  const AxisOff ox = axis_off(gx, g.lat_nx, 1, g.wrap[0]);
  const AxisOff oy = axis_off(gy, g.lat_ny, g.arr_nx, g.wrap[1]);
  const AxisOff oz = axis_off(gz, g.lat_nz, g.arr_nxy, g.wrap[2]);
  const size_t ds = g.dist_size;
  float f[L::Q];
  static_for<0, L::Q>([&](auto I) {
    if constexpr (PROP == PROP_AA_ODD) {
      const int off = dir_offset<L, I>(ox, oy, oz, false);
      f[I] = ld<NT>(p.din + ds * (size_t)L::opp(I) + (uint32_t)((int)gi + off));
    } else {
      f[I] = ld<NT>(p.din + ds * (size_t)I + gi);
    }
  });
  float rho, v[3];
  macro_standard<L, float>(f, p.cp.incompressible != 0, rho, v);
  check_invalid<float>(p.status, p.options, rho, gx, gy, gz);
  if (p.relaxation_enabled) {
    if constexpr (MODEL == 0) bgk_relax<L, float, FORCE>(f, rho, v, p.cp);
    else mrt_relax<L, float, FORCE>(f, v, p.cp, false);
  }
  if (p.options & 1u) {
    p.rho[gi] = rho;
    p.vx[gi] = v[0];
    p.vy[gi] = v[1];
    p.vz[gi] = v[2];
  }
  static_for<0, L::Q>([&](auto I) {
    if constexpr (PROP == PROP_AA_EVEN) {
      st<NT>(p.dout + ds * (size_t)L::opp(I) + gi, f[I]);
    } else {
      const int off = dir_offset<L, I>(ox, oy, oz, true);
      st<NT>(p.dout + ds * (size_t)I + (uint32_t)((int)gi + off), f[I]);
    }
  });
}


// Whole-row workgroup, aligned global accesses for the steps that stream along x
// (odd AA step: pull + push; AB: push).  Thread t owns the real node x = t + 1; every population row
// segment is loaded and stored at the thread's own x (one aligned 256-byte request per wave and
// direction); the +-1 x shift happens in registers: cross-lane shuffle inside a wave, a few LDS words
// between the waves of the row, and the periodic wrap as the cyclic continuation of the same exchange
// (x = 1 <-> x = nx).  This is the MI355X counterpart of the reference's shuffle / shared-memory
// propagation (propagation.mako:180-382).  Requires blockDim.x >= nx (the row fits one workgroup) and
// x wrapped in-sweep.
// NSEG = 2 (rows longer than 512 nodes): every thread owns TWO nodes, x = t + 1 and x = t + 1 + blockDim.x, i.e. the
// workgroup walks the row as two segments at once.  A 1024-node row is then 8 waves with 38 loads in flight each
// instead of 16 waves that meet at every barrier (4.7 -> 6 TB/s on the odd step, profiles/r02/README.md); the
// exchange is the same code with "virtual waves" vw = segment * waves + wave.
template <int MODEL, int PROP, bool FORCE, int NSEG>
__global__ void __launch_bounds__(512, (NSEG == 2 && MODEL == 0 && !FORCE) ? 6 : (NSEG == 1 ? 4 : 2))
fast_row_kernel(const SweepParams<D3Q19, float> p) {
  using L = D3Q19;
  constexpr int NW = 16;
  __shared__ float s_in_p[NW][5], s_in_m[NW][5], s_out_p[NW][5], s_out_m[NW][5];
  __shared__ float s_inw_p[5], s_inw_m[5], s_wrap_p[5], s_wrap_m[5];
  const Geometry& g = p.g;
  const int gy = sgpr(p.y0 + (int)blockIdx.y);
  const int gz = sgpr(p.z0 + (int)blockIdx.z);
  const int nx = g.lat_nx - 2;
  const int lane = (int)threadIdx.x & 63;
  const int nwave = (int)blockDim.x >> 6;                       // waves per segment
  int x[NSEG], vw[NSEG];
  bool live[NSEG];
  uint32_t xb[NSEG];                                            // per-lane address registers: byte offset of x in its row
  static_for<0, NSEG>([&](auto S) {
    x[S] = (int)threadIdx.x + 1 + S * (int)blockDim.x;
    vw[S] = ((int)threadIdx.x >> 6) + S * nwave;
    live[S] = x[S] <= nx;
    xb[S] = (uint32_t)(live[S] ? x[S] : 1) * 4u;                // idle lanes: an in-row address, never stored
  });
  const uint32_t row = sgpr((uint32_t)g.arr_nx * (uint32_t)gy + (uint32_t)g.arr_nxy * (uint32_t)gz);
  AxisOff oy = axis_off(gy, g.lat_ny, g.arr_nx, g.wrap[1]);
  AxisOff oz = axis_off(gz, g.lat_nz, g.arr_nxy, g.wrap[2]);
  oy.p = sgpr(oy.p); oy.m = sgpr(oy.m); oz.p = sgpr(oz.p); oz.m = sgpr(oz.m);
  const AxisOff ox0 = {0, 0};
  const size_t ds = g.dist_size;

  float f[NSEG][L::Q];
  if constexpr (PROP == PROP_AA_ODD) {
    // raw_i(x) = slot opp(i) at (x, y - e_y, z - e_z): the value node x + e_x will use as f_i
    static_for<0, L::Q>([&](auto I) {
      const int off = dir_offset<L, I>(ox0, oy, oz, false);
      const auto base = uniform_base(p.din + ds * (size_t)L::opp(I) + (uint32_t)((int)row + off));
      static_for<0, NSEG>([&](auto S) { f[S][I] = ldg<NT>(at_byte(base, xb[S])); });
    });
    static_for<0, NSEG>([&](auto S) {
      int kp = 0, km = 0;
      static_for<1, L::Q>([&](auto I) {
        if constexpr (L::ex(I) > 0) {
          if (lane == 63) s_in_p[vw[S]][kp] = f[S][I];
          if (x[S] == nx) s_inw_p[kp] = f[S][I];
          kp++;
        }
        if constexpr (L::ex(I) < 0) {
          if (lane == 0) s_in_m[vw[S]][km] = f[S][I];
          if (x[S] == 1) s_inw_m[km] = f[S][I];
          km++;
        }
      });
    });
    __syncthreads();
    static_for<0, NSEG>([&](auto S) {
      int kp = 0, km = 0;
      static_for<1, L::Q>([&](auto I) {
        // one DPP wave shift per direction (slf_rowpush.h: lane_shift1); the lane without a source takes the word the
        // neighbouring wave left in LDS (a wave-uniform slot: every lane reads the same address) -- the row's first
        // node the wrap word; x = nx need not be lane 63 (nx % 64 != 0)
        if constexpr (L::ex(I) > 0) {
          const float edge = (vw[S] == 0) ? s_inw_p[kp] : s_in_p[(vw[S] - 1) & (NW - 1)][kp];
          f[S][I] = lane_shift1<float, true>(edge, f[S][I]);
          kp++;
        }
        if constexpr (L::ex(I) < 0) {
          float t = lane_shift1<float, false>(s_in_m[(vw[S] + 1) & (NW - 1)][km], f[S][I]);
          if (x[S] == nx) t = s_inw_m[km];
          f[S][I] = t;
          km++;
        }
      });
    });
  } else {
    static_for<0, L::Q>([&](auto I) {
      const auto base = uniform_base(p.din + ds * (size_t)I + row);
      static_for<0, NSEG>([&](auto S) { f[S][I] = ldg<NT>(at_byte(base, xb[S])); });
    });
  }

  static_for<0, NSEG>([&](auto S) {
    float rho, v[3];
    macro_standard<L, float>(f[S], p.cp.incompressible != 0, rho, v);
    if (live[S]) check_invalid<float>(p.status, p.options, rho, x[S], gy, gz);
    if (p.relaxation_enabled) {
      if constexpr (MODEL == 0) bgk_relax<L, float, FORCE>(f[S], rho, v, p.cp);
      else mrt_relax<L, float, FORCE>(f[S], v, p.cp, false);
    }
    if ((p.options & 1u) && live[S]) {
      const uint32_t gi = row + (uint32_t)x[S];
      p.rho[gi] = rho;
      p.vx[gi] = v[0];
      p.vy[gi] = v[1];
      p.vz[gi] = v[2];
    }
  });

  // push: the value of node x travels to x + e_x; it is stored by the thread that owns the target x
  static_for<0, NSEG>([&](auto S) {
    int kp = 0, km = 0;
    static_for<1, L::Q>([&](auto I) {
      if constexpr (L::ex(I) > 0) {
        if (lane == 63) s_out_p[vw[S]][kp] = f[S][I];
        if (x[S] == nx) s_wrap_p[kp] = f[S][I];
        kp++;
      }
      if constexpr (L::ex(I) < 0) {
        if (lane == 0) s_out_m[vw[S]][km] = f[S][I];
        if (x[S] == 1) s_wrap_m[km] = f[S][I];
        km++;
      }
    });
  });
  __syncthreads();
  {
    int kp = 0, km = 0;
    static_for<0, L::Q>([&](auto I) {
      const int off = dir_offset<L, I>(ox0, oy, oz, true);
      const auto base = uniform_base(p.dout + ds * (size_t)I + (uint32_t)((int)row + off));
      static_for<0, NSEG>([&](auto S) {
        float t = f[S][I];
        if constexpr (L::ex(I) > 0) {
          const float edge = (vw[S] == 0) ? s_wrap_p[kp] : s_out_p[(vw[S] - 1) & (NW - 1)][kp];
          t = lane_shift1<float, true>(edge, f[S][I]);
        }
        if constexpr (L::ex(I) < 0) {
          t = lane_shift1<float, false>(s_out_m[(vw[S] + 1) & (NW - 1)][km], f[S][I]);
          if (x[S] == nx) t = s_wrap_m[km];
        }
        if (live[S]) stg<NT>(at_byte(base, xb[S]), t);
      });
      if constexpr (L::ex(I) > 0) kp++;
      if constexpr (L::ex(I) < 0) km++;
    });
  }
}

template <int MODEL, bool FORCE>
static bool launch_fast_model(Prop prop, const Geometry& g, const SweepParams<D3Q19, float>& p, int ny, int nz,
                              int block_x, hipStream_t s) {
  const int variant = g.variant;
  const int nx = g.lat_nx - 2;
  // 8-byte accesses need x = 1 on an 8-byte boundary of both arrays
  const bool vec_ok = (variant & 2) && (((uintptr_t)(p.din + 1)) % 8 == 0) && (((uintptr_t)(p.dout + 1)) % 8 == 0) &&
                      (g.arr_nx % 2 == 0) && (g.dist_size % 2 == 0);
  if (prop == PROP_AA_EVEN && vec_ok) {
    const int threads_needed = (nx + 1) / 2;
    int bx = ((threads_needed + 63) / 64) * 64;
    if (bx > block_x) bx = block_x;
    if (bx > 512) bx = 512;
    dim3 block(bx, 1, 1);
    dim3 grid((threads_needed + bx - 1) / bx, ny, nz);
    if (p.xsend[0] || p.xsend[1]) {
      SweepParams<D3Q19, float> q = p;
      q.row_mode = xcd_shift_for((unsigned)ny, grid.x) << 4;      // rows that share face-buffer lines: one XCD
      hipLaunchKernelGGL((fast_even_kernel<MODEL, 2, FORCE, true>), grid, block, 0, s, q);
    } else {
      hipLaunchKernelGGL((fast_even_kernel<MODEL, 2, FORCE, false>), grid, block, 0, s, p);
    }
    return true;
  }
  if (!g.wrap[0]) return false;                  // everything below wraps x in-sweep
  if ((variant & 8) && prop != PROP_AA_EVEN) {
    if (nx > 1024) return false;                 // long rows: segmented row kernel (slf_row.hip)
    dim3 grid(1, ny, nz);
    if (nx > 512) {                              // two nodes per thread
      dim3 block((((nx + 1) / 2 + 63) / 64) * 64, 1, 1);
      if (prop == PROP_AB) hipLaunchKernelGGL((fast_row_kernel<MODEL, PROP_AB, FORCE, 2>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((fast_row_kernel<MODEL, PROP_AA_ODD, FORCE, 2>), grid, block, 0, s, p);
      return true;
    }
    dim3 block(((nx + 63) / 64) * 64, 1, 1);
    if (prop == PROP_AB) hipLaunchKernelGGL((fast_row_kernel<MODEL, PROP_AB, FORCE, 1>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((fast_row_kernel<MODEL, PROP_AA_ODD, FORCE, 1>), grid, block, 0, s, p);
    return true;
  }
  dim3 block(block_x, 1, 1);
  dim3 grid((nx + block_x - 1) / block_x, ny, nz);
  switch (prop) {
    case PROP_AB: hipLaunchKernelGGL((fast_scalar_kernel<MODEL, PROP_AB, FORCE>), grid, block, 0, s, p); break;
    case PROP_AA_EVEN: hipLaunchKernelGGL((fast_scalar_kernel<MODEL, PROP_AA_EVEN, FORCE>), grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL((fast_scalar_kernel<MODEL, PROP_AA_ODD, FORCE>), grid, block, 0, s, p); break;
  }
  return true;
}

bool launch_sweep_fast(const KernelSelector& sel, Prop prop, const Geometry& g, const Physics& ph, const SweepArgs& a,
                       int y0, int y1, int z0, int z1, int block_x, hipStream_t s, hipError_t* err) {
  if (sel.general || sel.lattice != 1 || sel.precision != 4 || !(g.variant & 1)) return false;
  // x not wrapped in-sweep (ghost columns or x-face buffers): the even step is local to the node, its two-nodes-per-
  // thread form applies when no thread straddles the ghost column; the x-streaming steps go to slf_row.hip
  if (!g.wrap[0] && !(prop == PROP_AA_EVEN && (g.variant & 2) && (g.lat_nx - 2) % 2 == 0)) return false;
  if (g.variant & 256) return false;   // tests: route around these kernels, to the fluid-only instantiations of slf_row.hip
  if (y1 <= y0 || z1 <= z0) return false;
  const SweepParams<D3Q19, float> p = make_params<D3Q19, float>(g, ph, a, y0, z0);
  const int ny = y1 - y0, nz = z1 - z0;
  bool done;
  if (ph.has_force) {
    done = sel.model == 0 ? launch_fast_model<0, true>(prop, g, p, ny, nz, block_x, s)
                          : launch_fast_model<1, true>(prop, g, p, ny, nz, block_x, s);
  } else {   // no body force: straight-line collision, half the registers, twice the resident waves
    done = sel.model == 0 ? launch_fast_model<0, false>(prop, g, p, ny, nz, block_x, s)
                          : launch_fast_model<1, false>(prop, g, p, ny, nz, block_x, s);
  }
  if (done) *err = hipGetLastError();
  return done;
}

}  // namespace slf
