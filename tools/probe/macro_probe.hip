// Stand-alone probe (GPU box only; not part of libsailfish_hip.so): what bounds a ShanChenPrepareMacroFields-shaped pass --
// 2 x 19 population arrays read once, five node arrays written -- on this box?  Same launch shape as sc_macro_kernel
// (one workgroup per 256-node row), then one thing changed at a time.  Prints ms and TB/s per variant.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/macro_probe tools/probe/macro_probe.hip && tools/probe/macro_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct P {
  const float* a;      // lattice 0: 19 arrays, `stride` floats apart
  const float* b;      // lattice 1
  float* out[5];
  size_t stride;
  int nx, arr_nx, ny;  // node (x, y, z) lives at z * arr_nx * ny + y * arr_nx + x
};

template <int NT> __device__ __forceinline__ float ld(const float* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// NS: streams per lattice (19), LAT: lattices read (1 or 2), NT: non-temporal loads, ST: stores (0 none, 1 plain, 2 nt),
// SHIFT: odd-step layout (x-moving directions read at x -/+ 1)
template <int LAT, int NT, int ST, int SHIFT>
__global__ void __launch_bounds__(1024) macro_k(const P p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= p.nx) return;
  const size_t row = ((size_t)blockIdx.z * p.ny + blockIdx.y) * p.arr_nx + 32;
  float r0 = 0.f, r1 = 0.f, m[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < LAT; l++) {
    const float* base = l ? p.b : p.a;
    float f[19];
#pragma unroll
    for (int i = 0; i < 19; i++) {
      const int sh = SHIFT ? ((i % 3) - 1) : 0;
      f[i] = ld<NT>(base + (size_t)i * p.stride + row + x + sh);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 19; i++) s += f[i];
    if (l) r1 = s; else r0 = s;
    m[0] += f[1] - f[2] + f[7] - f[8];
    m[1] += f[3] - f[4] + f[9] - f[10];
    m[2] += f[5] - f[6] + f[11] - f[12];
  }
  const size_t gi = row + x;
  if (ST == 0) {
    if (r0 + r1 + m[0] + m[1] + m[2] == 123456.789f) p.out[0][gi] = r0;
  } else if (ST == 1) {
    p.out[0][gi] = r0; p.out[1][gi] = r1; p.out[2][gi] = m[0]; p.out[3][gi] = m[1]; p.out[4][gi] = m[2];
  } else {
    __builtin_nontemporal_store(r0, p.out[0] + gi); __builtin_nontemporal_store(r1, p.out[1] + gi);
    __builtin_nontemporal_store(m[0], p.out[2] + gi); __builtin_nontemporal_store(m[1], p.out[3] + gi);
    __builtin_nontemporal_store(m[2], p.out[4] + gi);
  }
}

// four nodes per lane: 16-byte loads, one 64-lane wave per 256-node row (aligned layout only)
template <int LAT, int NT>
__global__ void __launch_bounds__(256) macro4_k(const P p) {
  const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (x >= p.nx) return;
  const size_t row = ((size_t)blockIdx.z * p.ny + blockIdx.y) * p.arr_nx + 32;     // 16-byte aligned start
  f4 r[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0;
#pragma unroll
  for (int l = 0; l < LAT; l++) {
    const float* base = l ? p.b : p.a;
    f4 f[19];
#pragma unroll
    for (int i = 0; i < 19; i++) {
      const f4* q = (const f4*)(base + (size_t)i * p.stride + row + x);
      f[i] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int i = 0; i < 19; i++) r[l] += f[i];
    m0 += f[1] - f[2] + f[7] - f[8];
    m1 += f[3] - f[4] + f[9] - f[10];
    m2 += f[5] - f[6] + f[11] - f[12];
  }
  const size_t gi = row + x;
  *(f4*)(p.out[0] + gi) = r[0]; *(f4*)(p.out[1] + gi) = r[1]; *(f4*)(p.out[2] + gi) = m0; *(f4*)(p.out[3] + gi) = m1;
  *(f4*)(p.out[4] + gi) = m2;
}

// R rows per workgroup, one after the other (fewer, longer-lived workgroups)
template <int R>
__global__ void __launch_bounds__(1024) macro_rows_k(const P p) {
  const int x = threadIdx.x;
#pragma unroll 1
  for (int k = 0; k < R; k++) {
    const size_t row = ((size_t)blockIdx.z * p.ny + blockIdx.y * R + k) * p.arr_nx + 32;
    float r0 = 0.f, r1 = 0.f, m[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < 2; l++) {
      const float* base = l ? p.b : p.a;
      float f[19];
#pragma unroll
      for (int i = 0; i < 19; i++) f[i] = ld<1>(base + (size_t)i * p.stride + row + x);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 19; i++) s += f[i];
      if (l) r1 = s; else r0 = s;
      m[0] += f[1] - f[2] + f[7] - f[8];
      m[1] += f[3] - f[4] + f[9] - f[10];
      m[2] += f[5] - f[6] + f[11] - f[12];
    }
    const size_t gi = row + x;
    p.out[0][gi] = r0; p.out[1][gi] = r1; p.out[2][gi] = m[0]; p.out[3][gi] = m[1]; p.out[4][gi] = m[2];
  }
}

// NOUT of the five node arrays stored (how much does each written stream cost?)
template <int NOUT>
__global__ void __launch_bounds__(1024) macro_nout_k(const P p) {
  const int x = threadIdx.x;
  const size_t row = ((size_t)blockIdx.z * p.ny + blockIdx.y) * p.arr_nx + 32;
  float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < 2; l++) {
    const float* base = l ? p.b : p.a;
    float f[19];
#pragma unroll
    for (int i = 0; i < 19; i++) f[i] = ld<1>(base + (size_t)i * p.stride + row + x);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 19; i++) s += f[i];
    o[l] = s;
    o[2] += f[1] - f[2] + f[7] - f[8];
    o[3] += f[3] - f[4] + f[9] - f[10];
    o[4] += f[5] - f[6] + f[11] - f[12];
  }
  const size_t gi = row + x;
  float rest = 0.f;
#pragma unroll
  for (int k = NOUT; k < 5; k++) rest += o[k];
#pragma unroll
  for (int k = 0; k < NOUT; k++) p.out[k][gi] = o[k] + (k == 0 ? rest : 0.f);
  if (NOUT == 0 && rest == 123456.789f) p.out[0][gi] = rest;
}

// the five results of a row handed over through LDS and stored as whole 1 KiB rows with 16-byte stores (5 store
// instructions per workgroup instead of 20)
template <int NTS>
__global__ void __launch_bounds__(256) macro_lds_k(const P p) {
  __shared__ float sh[5][256];
  const int x = threadIdx.x;
  const size_t row = ((size_t)blockIdx.z * p.ny + blockIdx.y) * p.arr_nx + 32;
  float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < 2; l++) {
    const float* base = l ? p.b : p.a;
    float f[19];
#pragma unroll
    for (int i = 0; i < 19; i++) f[i] = ld<1>(base + (size_t)i * p.stride + row + x);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 19; i++) s += f[i];
    o[l] = s;
    o[2] += f[1] - f[2] + f[7] - f[8];
    o[3] += f[3] - f[4] + f[9] - f[10];
    o[4] += f[5] - f[6] + f[11] - f[12];
  }
#pragma unroll
  for (int k = 0; k < 5; k++) sh[k][x] = o[k];
  __syncthreads();
  const int w = x >> 6, l = x & 63;
  for (int k = w; k < 5; k += 4) {
    const f4 v = *(const f4*)&sh[k][4 * l];
    f4* q = (f4*)(p.out[k] + row + 4 * l);
    if (NTS) __builtin_nontemporal_store(v, q); else *q = v;
  }
}

// NR arrays read, one written, 16 bytes per lane, same index everywhere: the read : write ratio by itself
template <int NR, int NT>
__global__ void __launch_bounds__(256) ratio_k(const P p, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f4 s = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const f4* q = (const f4*)(p.a + (size_t)k * p.stride) + i;
    s += NT ? __builtin_nontemporal_load(q) : *q;
  }
  f4* d = (f4*)p.out[0] + i;
  if (NT) __builtin_nontemporal_store(s, d); else *d = s;
}

// NR population arrays read (lattice 0 first, then lattice 1), NW arrays of a third set written, one dword per lane,
// one workgroup per row: which combinations of open read and write streams run below the plain-copy rate?
struct P2 { const float* a; const float* b; float* c; size_t stride; int nx, arr_nx, ny; };
template <int NR, int NW, int NT>
__global__ void __launch_bounds__(1024) rw_k(const P2 p) {
  const int x = threadIdx.x;
  const size_t row = ((size_t)blockIdx.z * p.ny + blockIdx.y) * p.arr_nx + 32;
  float f[NR];
#pragma unroll
  for (int i = 0; i < NR; i++) {
    const float* q = (i < 19 ? p.a + (size_t)i * p.stride : p.b + (size_t)(i - 19) * p.stride) + row + x;
    f[i] = NT ? __builtin_nontemporal_load(q) : *q;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NR; i++) s += f[i];
  if (NW == 0) {
    if (s == 123456.789f) p.c[row + x] = s;
    return;
  }
#pragma unroll
  for (int k = 0; k < NW; k++) {
    const float v = s + f[k % NR];
    float* q = p.c + (size_t)k * p.stride + row + x;
    if (NT) __builtin_nontemporal_store(v, q); else *q = v;
  }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

template <class F>
static void time_it(const char* name, double bytes, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; i++) launch();
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int rep = 0; rep < 7; rep++) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms / 20);
  }
  CK(hipGetLastError());
  std::sort(t.begin(), t.end());
  printf("%-90s median %.4f ms  %.2f TB/s   (min %.4f max %.4f)\n", name, t[3], bytes / t[3] * 1e-9, t[0], t[6]);
  fflush(stdout);
}

// mode 0 (default): separate hipMalloc allocations, every variant.  mode 1: one 200 GiB arena, the arrays put at chosen
// offsets (GiB) -- does it matter where the five written arrays lie relative to the populations and to each other?
static void run_set(const char* tag, const P& p, int n, int nz, bool all) {
  const double N = (double)n * n * nz;
  dim3 grid(1, n, nz), block(n);
  char name[160];
#define T(label, bytes, ...) do { snprintf(name, sizeof name, "%s %s", tag, label); time_it(name, bytes, [&] { __VA_ARGS__; }); } while (0)
  T("A  nt loads, 5 plain stores (as sc_macro_kernel)", N * 172, hipLaunchKernelGGL((macro_k<2, 1, 1, 0>), grid, block, 0, 0, p));
  T("B  plain loads, plain stores", N * 172, hipLaunchKernelGGL((macro_k<2, 0, 1, 0>), grid, block, 0, 0, p));
  T("C  nt loads, nt stores", N * 172, hipLaunchKernelGGL((macro_k<2, 1, 2, 0>), grid, block, 0, 0, p));
  T("C' plain loads, nt stores", N * 172, hipLaunchKernelGGL((macro_k<2, 0, 2, 0>), grid, block, 0, 0, p));
  T("D  nt loads, no stores", N * 152, hipLaunchKernelGGL((macro_k<2, 1, 0, 0>), grid, block, 0, 0, p));
  T("D' plain loads, no stores", N * 152, hipLaunchKernelGGL((macro_k<2, 0, 0, 0>), grid, block, 0, 0, p));
  if (!all) return;
  T("E  one lattice: 19 streams, no stores", N * 76, hipLaunchKernelGGL((macro_k<1, 1, 0, 0>), grid, block, 0, 0, p));
  T("F  as A, x-shifted reads (odd-step layout)", N * 172, hipLaunchKernelGGL((macro_k<2, 1, 1, 1>), grid, block, 0, 0, p));
  T("G  as A, 64-thread workgroups", N * 172, hipLaunchKernelGGL((macro_k<2, 1, 1, 0>), dim3(n / 64, n, nz), dim3(64), 0, 0, p));
  T("H  16-byte nt loads, one wave per row", N * 172, hipLaunchKernelGGL((macro4_k<2, 1>), dim3(1, n, nz), dim3(n / 4), 0, 0, p));
  T("I  16-byte plain loads", N * 172, hipLaunchKernelGGL((macro4_k<2, 0>), dim3(1, n, nz), dim3(n / 4), 0, 0, p));
  T("N0 nt loads, 0 of 5 arrays stored", N * 152, hipLaunchKernelGGL((macro_nout_k<0>), grid, block, 0, 0, p));
  T("N1 nt loads, 1 of 5 arrays stored", N * 156, hipLaunchKernelGGL((macro_nout_k<1>), grid, block, 0, 0, p));
  T("N2 nt loads, 2 of 5 arrays stored", N * 160, hipLaunchKernelGGL((macro_nout_k<2>), grid, block, 0, 0, p));
  T("N3 nt loads, 3 of 5 arrays stored", N * 164, hipLaunchKernelGGL((macro_nout_k<3>), grid, block, 0, 0, p));
  T("N5 nt loads, 5 of 5 arrays stored", N * 172, hipLaunchKernelGGL((macro_nout_k<5>), grid, block, 0, 0, p));
  T("L  results through LDS, 1 KiB row stores", N * 172, hipLaunchKernelGGL((macro_lds_k<0>), grid, block, 0, 0, p));
  T("L' same, nt stores", N * 172, hipLaunchKernelGGL((macro_lds_k<1>), grid, block, 0, 0, p));
  { const size_t n4 = (size_t)p.stride / 4;
    T("R8  8 arrays read : 1 written, 16 B per lane, nt", n4 * 16.0 * 9, hipLaunchKernelGGL((ratio_k<8, 1>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, p, n4));
    T("R8' same, plain", n4 * 16.0 * 9, hipLaunchKernelGGL((ratio_k<8, 0>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, p, n4));
    T("R1  1 array read : 1 written, nt", n4 * 16.0 * 2, hipLaunchKernelGGL((ratio_k<1, 1>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, p, n4));
    T("R19 19 arrays read : 1 written, nt", n4 * 16.0 * 20, hipLaunchKernelGGL((ratio_k<19, 1>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, p, n4)); }
  T("J  4 rows per workgroup in turn", N * 172, hipLaunchKernelGGL((macro_rows_k<4>), dim3(1, n / 4, nz), block, 0, 0, p));
#undef T
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  P p;
  p.nx = n; p.ny = n; p.arr_nx = n + 64;
  const int nz = n;
  const size_t nodes = (size_t)p.arr_nx * n * nz + 256;
  p.stride = nodes;
  printf("box %d^3, arr_nx %d, %.2f GB of populations, %.3f GB per node array\n", n, p.arr_nx, nodes * 38 * 4e-9, nodes * 4e-9);
  if (mode == 0) {
    float *a, *b;
    CK(hipMalloc(&a, nodes * 19 * 4)); CK(hipMalloc(&b, nodes * 19 * 4));
    CK(hipMemset(a, 0, nodes * 19 * 4)); CK(hipMemset(b, 0, nodes * 19 * 4));
    p.a = a; p.b = b;
    for (int i = 0; i < 5; i++) { CK(hipMalloc(&p.out[i], nodes * 4)); CK(hipMemset(p.out[i], 0, nodes * 4)); }
    run_set("malloc", p, n, nz, true);
    return 0;
  }
  if (mode == 2) {
    P2 q;
    q.nx = n; q.ny = n; q.arr_nx = p.arr_nx; q.stride = nodes;
    float *a, *b, *c;
    CK(hipMalloc(&a, nodes * 19 * 4)); CK(hipMalloc(&b, nodes * 19 * 4)); CK(hipMalloc(&c, nodes * 38 * 4));
    CK(hipMemset(a, 0, nodes * 19 * 4)); CK(hipMemset(b, 0, nodes * 19 * 4)); CK(hipMemset(c, 0, nodes * 38 * 4));
    q.a = a; q.b = b; q.c = c;
    const double N = (double)n * n * nz;
    dim3 grid(1, n, nz), block(n);
    char name[100];
#define RW(NR, NW, NT) do { snprintf(name, sizeof name, "%2d read, %2d written, %s", NR, NW, NT ? "nt" : "plain"); \
    time_it(name, N * 4 * (NR + NW), [&] { hipLaunchKernelGGL((rw_k<NR, NW, NT>), grid, block, 0, 0, q); }); } while (0)
    RW(8, 0, 1); RW(8, 1, 1); RW(8, 5, 1); RW(8, 8, 1);
    RW(19, 0, 1); RW(19, 1, 1); RW(19, 5, 1); RW(19, 19, 1);
    RW(38, 0, 1); RW(38, 1, 1); RW(38, 5, 1); RW(38, 19, 1); RW(38, 38, 1);
    RW(19, 5, 0); RW(19, 19, 0); RW(38, 5, 0); RW(38, 38, 0);
    return 0;
  }
  const size_t GiB = 1ull << 30;
  char* arena;
  CK(hipMalloc(&arena, 200 * GiB));
  struct Layout { const char* tag; double a, b, out[5]; double slot_gib; };    // offsets in GiB; slot_gib: distance of the 19 arrays (0 = packed)
  const Layout layouts[] = {
    {"packed          ", 0, 1.7, {3.4, 3.5, 3.6, 3.7, 3.8}, 0},
    {"outs far, packed", 0, 1.7, {100, 100.1, 100.2, 100.3, 100.4}, 0},
    {"outs 20 GiB apart", 0, 1.7, {60, 80, 100, 120, 140}, 0},
    {"a|b 40 apart, outs 20 apart", 0, 40, {80, 100, 120, 140, 160}, 0},
    {"slots 4 GiB apart, outs packed", 0, 2, {190, 190.1, 190.2, 190.3, 190.4}, 4},
    {"slots 4 GiB apart, outs spread", 0, 2, {1, 41, 81, 121, 161}, 4},
    {"slots 4 GiB apart, outs next to slots", 0, 2, {1, 5, 9, 13, 17}, 4},
  };
  for (const Layout& L : layouts) {
    p.stride = L.slot_gib > 0 ? (size_t)(L.slot_gib * GiB) / 4 : nodes;
    p.a = (const float*)(arena + (size_t)(L.a * GiB));
    p.b = (const float*)(arena + (size_t)(L.b * GiB));
    for (int i = 0; i < 19; i++) {
      CK(hipMemset((void*)(p.a + i * p.stride), 0, nodes * 4));
      CK(hipMemset((void*)(p.b + i * p.stride), 0, nodes * 4));
    }
    for (int i = 0; i < 5; i++) { p.out[i] = (float*)(arena + (size_t)(L.out[i] * GiB)); CK(hipMemset(p.out[i], 0, nodes * 4)); }
    run_set(L.tag, p, n, nz, false);
  }
  return 0;
}
