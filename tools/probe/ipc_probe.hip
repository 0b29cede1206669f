// Stand-alone probe (GPU box only): what a device-side halo transport between the PROCESSES of one node can rely on.
// N processes (fork before any HIP call) on ONE device:
//   1. hipIpcGetMemHandle / hipIpcOpenMemHandle of a plain hipMalloc buffer and of an uncached (hipDeviceMallocUncached)
//      flag block, across processes that share the device;
//   2. ring ping: process r stores into the flag block of r + 1 (mapped) with a one-lane kernel, r + 1 spins on its own
//      flag with a one-lane kernel -- round-trip time, forward progress with N spinning processes;
//   3. coherence of PLAIN (cacheable) memory written through the mapping: writer kernel fills the neighbour's buffer,
//      event record, signal kernel; the neighbour's wait kernel, then a checking kernel -- R rounds, patterns change;
//   4. aggregate copy bandwidth of N processes running a copy kernel at the same time against one process alone.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/ipc_probe tools/probe/ipc_probe.hip && tools/probe/ipc_probe 8
// Every spin is bounded (wall_clock64, 100 MHz): a lost signal prints TIMEOUT instead of hanging the box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <time.h>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("[%d] HIP error %s line %d\n", g_rank, hipGetErrorString(_e), __LINE__); fflush(stdout); _exit(2); } } while (0)
static int g_rank = -1;

typedef unsigned long long u64;

__global__ void signal_k(u64* flag, u64 value) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// waits until *flag >= value; timeout in 100 MHz ticks; *err = 1 on timeout
__global__ void wait_k(const u64* flag, u64 value, u64 timeout, int* err) {
  if (threadIdx.x != 0) return;
  const u64 t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > timeout) { *err = 1; return; }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

__global__ void fill_k(unsigned* dst, size_t n, unsigned pattern) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = pattern ^ (unsigned)i;
}

// scattered 4-byte stores, one lane per 'row' (the x-face pattern): 32 rows share a 128-byte line
__global__ void fill_sparse_k(unsigned* dst, size_t n, unsigned pattern) {
  const size_t row = (size_t)blockIdx.x;
  if (threadIdx.x == 0 && row < n) dst[row] = pattern ^ (unsigned)row;
}

__global__ void check_k(const unsigned* src, size_t n, unsigned pattern, unsigned* bad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && src[i] != (pattern ^ (unsigned)i)) atomicAdd(bad, 1u);
}

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) copy_k(f4* __restrict__ dst, const f4* __restrict__ src, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

struct Shared {
  hipIpcMemHandle_t buf[64], flag[64];
  volatile int stage[64];
  double out[64][8];
};

static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void barrier(Shared* sh, int n, int rank, int stage) {
  sh->stage[rank] = stage;
  const double t0 = now();
  for (int i = 0; i < n; i++)
    while (sh->stage[i] < stage) { usleep(200); if (now() - t0 > 120) { printf("[%d] barrier %d timed out on %d\n", rank, stage, i); fflush(stdout); _exit(3); } }
}

static int child(Shared* sh, int n, int rank, size_t nwords, int rounds, size_t copy_mb) {
  g_rank = rank;
  CK(hipSetDevice(0));
  unsigned *buf = nullptr, *bad = nullptr;
  u64* flags = nullptr;
  int* err = nullptr;
  CK(hipMalloc(&buf, nwords * 4));
  CK(hipMemset(buf, 0, nwords * 4));
  hipError_t e = hipExtMallocWithFlags((void**)&flags, 4096, hipDeviceMallocUncached);
  if (e != hipSuccess) { printf("[%d] uncached alloc: %s; fine-grained instead\n", rank, hipGetErrorString(e)); CK(hipExtMallocWithFlags((void**)&flags, 4096, hipDeviceMallocFinegrained)); }
  CK(hipMemset(flags, 0, 4096));
  CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  CK(hipHostMalloc(&err, 4, hipHostMallocMapped)); *err = 0;
  CK(hipIpcGetMemHandle(&sh->buf[rank], buf));
  CK(hipIpcGetMemHandle(&sh->flag[rank], flags));
  CK(hipDeviceSynchronize());
  barrier(sh, n, rank, 1);
  const int up = (rank + 1) % n;
  unsigned* nbuf = buf; u64* nflags = flags;
  if (up != rank) {
    CK(hipIpcOpenMemHandle((void**)&nbuf, sh->buf[up], hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle((void**)&nflags, sh->flag[up], hipIpcMemLazyEnablePeerAccess));
  }
  if (rank == 0) printf("ipc open ok (%d processes)\n", n);
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const u64 TO = 10ull * 100000000ull;     // 10 s
  barrier(sh, n, rank, 2);

  // 2. ring ping: a token goes round the ring `rounds` times; everybody else spins meanwhile
  double t0 = now();
  for (int r = 1; r <= rounds; r++) {
    if (rank == 0) {
      signal_k<<<1, 64, 0, s>>>(nflags + 0, (u64)r);
      wait_k<<<1, 64, 0, s>>>(flags + 0, (u64)r, TO, err);
    } else {
      wait_k<<<1, 64, 0, s>>>(flags + 0, (u64)r, TO, err);
      signal_k<<<1, 64, 0, s>>>(nflags + 0, (u64)r);
    }
  }
  CK(hipStreamSynchronize(s));
  double dt = now() - t0;
  sh->out[rank][0] = dt / rounds * 1e6;
  if (*err) { printf("[%d] TIMEOUT in ring ping\n", rank); fflush(stdout); _exit(4); }
  barrier(sh, n, rank, 3);
  if (rank == 0) printf("ring ping: %.1f us per lap of %d hops = %.1f us per hop (all %d processes spinning)\n", sh->out[0][0], n, sh->out[0][0] / n, n);

  // 3. coherence of cacheable memory written through the mapping, dense and scattered stores, with a reader that has
  //    read the old contents just before (stale lines in its caches if anything keeps them)
  unsigned total_bad = 0;
  for (int sparse = 0; sparse < 2; sparse++) {
    for (int r = 1; r <= rounds; r++) {
      const unsigned pat = 0x9e3779b9u * (unsigned)(r + 1000 * sparse) + (unsigned)rank;
      const unsigned pat_in = 0x9e3779b9u * (unsigned)(r + 1000 * sparse) + (unsigned)((rank + n - 1) % n);
      const u64 seq = (u64)(sparse * rounds + r);
      // flag 1: "written" (me -> up); flag 2: "read, you may overwrite" (me -> down, stored in down's block... kept simple:
      // the writer waits for the reader's ack of the previous round on ITS OWN flag 2, which the reader (up) sets through
      // a second mapping -- here the ring is closed the other way round through process-shared host memory instead)
      if (sparse) fill_sparse_k<<<(unsigned)nwords, 64, 0, s>>>(nbuf, nwords, pat);
      else fill_k<<<(unsigned)((nwords + 255) / 256), 256, 0, s>>>(nbuf, nwords, pat);
      CK(hipEventRecord(ev, s));
      CK(hipStreamWaitEvent(s2, ev, 0));
      signal_k<<<1, 64, 0, s2>>>(nflags + 8, seq);
      wait_k<<<1, 64, 0, s2>>>(flags + 8, seq, TO, err);
      check_k<<<(unsigned)((nwords + 255) / 256), 256, 0, s2>>>(buf, nwords, pat_in, bad);
      CK(hipStreamSynchronize(s2));
      CK(hipStreamSynchronize(s));
      if (*err) { printf("[%d] TIMEOUT in coherence round %d\n", rank, r); fflush(stdout); _exit(4); }
      barrier(sh, n, rank, 100 + sparse * rounds * 2 + r);      // WAR: everybody has checked before anybody overwrites
    }
    unsigned h = 0; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    total_bad += h;
  }
  sh->out[rank][1] = total_bad;
  barrier(sh, n, rank, 100000);
  if (rank == 0) { double b = 0; for (int i = 0; i < n; i++) b += sh->out[i][1]; printf("coherence: %d rounds dense + %d scattered, %zu words: %.0f bad words over all processes\n", rounds, rounds, nwords, b); }

  // 4. copy bandwidth: everybody at once, then rank 0 alone
  const size_t n4 = copy_mb * (1 << 20) / 16;
  f4 *a, *b2;
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b2, n4 * 16));
  CK(hipMemset(a, 1, n4 * 16));
  CK(hipDeviceSynchronize());
  for (int phase = 0; phase < 2; phase++) {
    barrier(sh, n, rank, 100001 + 2 * phase);
    if (phase == 0 || rank == 0) {
      for (int i = 0; i < 3; i++) copy_k<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>(b2, a, n4);
      CK(hipStreamSynchronize(s));
    }
    barrier(sh, n, rank, 100002 + 2 * phase);
    if (phase == 0 || rank == 0) {
      t0 = now();
      const int reps = 20;
      for (int i = 0; i < reps; i++) copy_k<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>(b2, a, n4);
      CK(hipStreamSynchronize(s));
      dt = now() - t0;
      sh->out[rank][2 + phase] = 2.0 * n4 * 16 * reps / dt / 1e9;
    }
  }
  barrier(sh, n, rank, 100010);
  if (rank == 0) {
    double sum = 0; for (int i = 0; i < n; i++) sum += sh->out[i][2];
    printf("copy %zu MiB: %d processes at once %.0f GB/s in total (", copy_mb, n, sum);
    for (int i = 0; i < n; i++) printf("%.0f ", sh->out[i][2]);
    printf("), one process alone %.0f GB/s\n", sh->out[0][3]);
  }
  fflush(stdout);
  if (up != rank) { CK(hipIpcCloseMemHandle(nbuf)); CK(hipIpcCloseMemHandle(nflags)); }
  barrier(sh, n, rank, 100020);
  return 0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2;
  const size_t nwords = argc > 2 ? (size_t)atol(argv[2]) : (size_t)1 << 18;
  const int rounds = argc > 3 ? atoi(argv[3]) : 50;
  const size_t copy_mb = argc > 4 ? (size_t)atol(argv[4]) : 512;
  Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset(sh, 0, sizeof(Shared));
  std::vector<pid_t> kids;
  for (int r = 0; r < n; r++) {
    pid_t p = fork();
    if (p == 0) _exit(child(sh, n, r, nwords, rounds, copy_mb));
    kids.push_back(p);
  }
  int rc = 0;
  for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1; }
  printf("ipc_probe %d processes: %s\n", n, rc ? "FAILED" : "ok");
  return rc;
}
