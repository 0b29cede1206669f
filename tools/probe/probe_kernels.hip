// Box-calibration kernels for tools/box_probe.py (measurement tooling, not part of libsailfish_hip.so):
// plain streaming copy / read / write with 16-byte accesses, with and without the non-temporal hint,
// so that the sweep's HBM rate can be quoted against what THIS box's memory system delivers to the
// simplest possible kernel in the same process at the same moment (the guide's 6.29 TB/s "float4 copy").
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ void __launch_bounds__(256) copy_k(f4* __restrict__ dst, const f4* __restrict__ src, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f4 v = (NT & 1) ? __builtin_nontemporal_load(src + i) : src[i];
  if (NT & 2) __builtin_nontemporal_store(v, dst + i);
  else dst[i] = v;
}

template <int NT>
__global__ void __launch_bounds__(256) read_k(float* __restrict__ sink, const f4* __restrict__ src, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f4 v = (NT & 1) ? __builtin_nontemporal_load(src + i) : src[i];
  if (v[0] + v[1] + v[2] + v[3] == 123456.789f) sink[0] = v[0];   // never true for the probe data
}

template <int NT>
__global__ void __launch_bounds__(256) write_k(f4* __restrict__ dst, size_t n4, float val) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f4 v = {val, val, val, val};
  if (NT & 2) __builtin_nontemporal_store(v, dst + i);
  else dst[i] = v;
}

extern "C" {
// kind: 0 copy, 1 read, 2 write; nt: bit 0 loads, bit 1 stores
int probe_launch(int kind, int nt, void* dst, const void* src, size_t bytes, void* stream) {
  const size_t n4 = bytes / 16;
  const unsigned blocks = (unsigned)((n4 + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (kind == 0) {
    if (nt == 3) hipLaunchKernelGGL(copy_k<3>, dim3(blocks), dim3(256), 0, s, (f4*)dst, (const f4*)src, n4);
    else hipLaunchKernelGGL(copy_k<0>, dim3(blocks), dim3(256), 0, s, (f4*)dst, (const f4*)src, n4);
  } else if (kind == 1) {
    if (nt) hipLaunchKernelGGL(read_k<1>, dim3(blocks), dim3(256), 0, s, (float*)dst, (const f4*)src, n4);
    else hipLaunchKernelGGL(read_k<0>, dim3(blocks), dim3(256), 0, s, (float*)dst, (const f4*)src, n4);
  } else {
    if (nt) hipLaunchKernelGGL(write_k<2>, dim3(blocks), dim3(256), 0, s, (f4*)dst, n4, 1.0f);
    else hipLaunchKernelGGL(write_k<0>, dim3(blocks), dim3(256), 0, s, (f4*)dst, n4, 1.0f);
  }
  return (int)hipGetLastError();
}
}

// K streams S bytes apart: block b works on stream b % K, chunk b / K (4 KiB chunks).  kind: 0 write, 1 read,
// 2 in-place read-modify-write.  Emulates "K arrays written concurrently" with a controlled separation.
template <int KIND>
__global__ void __launch_bounds__(256) kstream_k(f4* __restrict__ base, size_t chunks_per_stream, int K, size_t s_bytes,
                                                 float* sink) {
  const size_t b = blockIdx.x;
  const size_t stream = b % (size_t)K, chunk = b / (size_t)K;
  if (chunk >= chunks_per_stream) return;
  f4* p = (f4*)((char*)base + stream * s_bytes + chunk * 4096) + threadIdx.x;
  if (KIND == 0) {
    f4 v = {1.f, 2.f, 3.f, 4.f};
    __builtin_nontemporal_store(v, p);
  } else if (KIND == 1) {
    f4 v = __builtin_nontemporal_load(p);
    if (v[0] + v[1] + v[2] + v[3] == 123456.789f) sink[0] = v[0];
  } else {
    f4 v = __builtin_nontemporal_load(p);
    v[0] += 1.f;
    __builtin_nontemporal_store(v, p);
  }
}

extern "C" int probe_kstream(int kind, void* base, size_t bytes_per_stream, int K, size_t s_bytes, void* sink, void* stream) {
  const size_t chunks = bytes_per_stream / 4096;
  const size_t blocks = chunks * (size_t)K;
  if (blocks > 0x7fffffffull) return -1;
  hipStream_t s = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL(kstream_k<0>, dim3((unsigned)blocks), dim3(256), 0, s, (f4*)base, chunks, K, s_bytes, (float*)sink);
  else if (kind == 1) hipLaunchKernelGGL(kstream_k<1>, dim3((unsigned)blocks), dim3(256), 0, s, (f4*)base, chunks, K, s_bytes, (float*)sink);
  else hipLaunchKernelGGL(kstream_k<2>, dim3((unsigned)blocks), dim3(256), 0, s, (f4*)base, chunks, K, s_bytes, (float*)sink);
  return (int)hipGetLastError();
}
