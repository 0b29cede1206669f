// Stand-alone probe (GPU box only): can a copy-shaped pass beat the 6.3 TB/s of a plain copy by showing HBM long
// read-only and write-only phases (7.1 / 6.8 TB/s each on this part) instead of a mixed stream?  The data has to wait
// somewhere between the phases; the only place big enough is the 256 MB memory-side cache: pass 1 reads a chunk of the
// source and stores it into a staging buffer small enough to stay there, pass 2 moves the staging buffer to the
// destination.  Prints the plain copy, the two passes by themselves and the pair, for several staging sizes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/phase_probe tools/probe/phase_probe.hip && tools/probe/phase_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

// LN: non-temporal loads, SN: non-temporal stores
template <int LN, int SN>
__global__ void __launch_bounds__(256) copy_k(f4* __restrict__ dst, const f4* __restrict__ src, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f4 v = LN ? __builtin_nontemporal_load(src + i) : src[i];
  if (SN) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

template <class F>
static double time_it(F run) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  run(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int rep = 0; rep < 5; rep++) {
    CK(hipEventRecord(e0, 0));
    run();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms);
  }
  CK(hipGetLastError());
  std::sort(t.begin(), t.end());
  return t[2];
}

template <int LN, int SN>
static void launch(void* d, const void* s, size_t bytes) {
  const size_t n4 = bytes / 16;
  hipLaunchKernelGGL((copy_k<LN, SN>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (f4*)d, (const f4*)s, n4);
}

int main() {
  const size_t GiB = 1ull << 30, MiB = 1ull << 20;
  const size_t total = 8 * GiB;
  char *src, *dst, *stage;
  CK(hipMalloc(&src, total)); CK(hipMalloc(&dst, total)); CK(hipMalloc(&stage, 512 * MiB));
  CK(hipMemset(src, 1, total)); CK(hipMemset(dst, 0, total)); CK(hipMemset(stage, 0, 512 * MiB));
  double ms = time_it([&] { launch<1, 1>(dst, src, total); });
  printf("plain copy, non-temporal both ways      %8.3f ms  %.2f TB/s (of 2 x 8 GiB)\n", ms, 2.0 * total / ms * 1e-9);
  ms = time_it([&] { launch<0, 0>(dst, src, total); });
  printf("plain copy, default policy              %8.3f ms  %.2f TB/s\n", ms, 2.0 * total / ms * 1e-9);
  for (size_t smib : {32, 64, 96, 128, 192, 384}) {
    const size_t S = smib * MiB;
    const int chunks = (int)(total / S);
    const double p1 = time_it([&] { for (int c = 0; c < chunks; c++) launch<1, 0>(stage, src + (size_t)c * S, S); });
    const double p2 = time_it([&] { for (int c = 0; c < chunks; c++) launch<0, 1>(dst + (size_t)c * S, stage, S); });
    const double pr = time_it([&] { for (int c = 0; c < chunks; c++) { launch<1, 0>(stage, src + (size_t)c * S, S); launch<0, 1>(dst + (size_t)c * S, stage, S); } });
    const double pn = time_it([&] { for (int c = 0; c < chunks; c++) { launch<1, 1>(stage, src + (size_t)c * S, S); launch<1, 1>(dst + (size_t)c * S, stage, S); } });
    printf("staging %3zu MiB x %3d: source -> staging %7.3f ms (%.2f TB/s of the 8 GiB read) | staging -> destination %7.3f ms (%.2f TB/s of the 8 GiB written) | pair %7.3f ms = %.2f TB/s as a copy | pair, all non-temporal %7.3f ms\n",
           smib, chunks, p1, total / p1 * 1e-9, p2, total / p2 * 1e-9, pr, 2.0 * total / pr * 1e-9, pn);
    fflush(stdout);
  }
  return 0;
}
