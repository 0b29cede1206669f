// C ABI of libsailfish_hip.so (see include/sailfish_hip.h for the contract and
// the reference interfaces each entry point replaces).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/sailfish_hip.h"
#include "slf_kernels.h"
#include "slf_node.h"

#ifndef SLF_DEFAULT_VARIANT
#define SLF_DEFAULT_VARIANT 11
#endif

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  return fail(SLF_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define SLF_HIP(call)                            \
  do {                                           \
    hipError_t _e = (call);                      \
    if (_e != hipSuccess) return hip_fail(_e, #call); \
  } while (0)

enum KernelKind {
  KK_COLLIDE_AND_PROPAGATE,
  KK_SET_INITIAL_CONDITIONS,
  KK_PBC,
  KK_PBC_SWAP,
  KK_MACRO_PBC,
  KK_COLLECT_SPARSE,
  KK_DISTRIBUTE_SPARSE,
  KK_COLLECT_BOX,
  KK_DISTRIBUTE_BOX,
  KK_COLLECT_FACE_SWAP,       // reference argument lists (kernel_utils.mako): ...ContinuousDataWithSwap
  KK_DISTRIBUTE_FACE_SWAP,
  KK_COLLECT_MACRO_FACE,      // ...ContinuousMacroData
  KK_DISTRIBUTE_MACRO_FACE,
  KK_COMPUTE_MACRO,
  KK_SC_MACRO,
  KK_SC_SWEEP0,
  KK_SC_SWEEP1,
  KK_SC_FUSED,
  KK_SC_INIT,
  KK_SCS_MACRO,
  KK_SCS_SWEEP,
  KK_RESIDENT,                // several steps inside one launch (slf_resident.hip)
};

}  // namespace

struct slf_ctx {
  int device;
};
struct slf_stream {
  slf_ctx* ctx;
  hipStream_t s;
};
struct slf_event {
  slf_ctx* ctx;
  hipEvent_t e;
};
struct slf_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};
struct slf_module {
  slf_ctx* ctx;
  slf::KernelSelector sel;
  slf::Geometry geo;
  slf::Physics phys;
  slf::ShanChen sc;
  int access_pattern;
  int block_x;
  void* node_params;  // device copy, in the module's precision
  uint32_t* status;   // device {flag, x, y, z} of the on-GPU invalid value check
  void* xsend[2];     // x-face buffers (slf_module_set_xface_buffers), NULL = unused
  void* xrecv[2];
  void* sc_send[3][2];  // binary Shan-Chen over connected x faces (slf_module_set_xface_planes): [lattice 0, lattice 1,
  void* sc_recv[3][2];  // densities][low / high face], NULL = unused
  slf::RowClasses rows;   // slf_module_classify_rows; rows.map == NULL: nothing classified
  void* rows_mem;         // one device allocation behind the tables of `rows`
  slf::SlotTable slots;   // indirect addressing: slot -> node, built from the `nodes` table of the first launch that names it
  void* slots_mem;
  int n_node_params, precision;
  // slf_module_update_node_params: pinned staging buffers for long updates, each with the event of its last copy
  void* stage[4];
  hipEvent_t stage_ev[4];
  size_t stage_bytes[4];
  int stage_next;
};
struct slf_kernel {
  slf_module* mod;
  KernelKind kind;
  std::vector<unsigned long long> ptrs;  // 'P' args in order
  std::vector<long long> ints;           // 'i' args in order
  int needs_iteration;
  uint32_t iteration;
  bool bound;
  bool sc_local_velocity;   // ShanChenPrepareDensities / ShanChenCollideAndPropagateFusedV
};

static hipStream_t native(slf_stream* s) { return s ? s->s : (hipStream_t)0; }

// Indirect addressing: the slot -> node table of `nodes` (SlotTable, slf_kernels.h), built on first use and whenever a
// launch names another table.  SLF_INDIRECT_SLOTS=0: never (one thread per dense node, the round-3 kernels: A/B switch).
static const slf::SlotTable* slot_table_for(slf_module* m, const void* nodes, hipStream_t s) {
  static const bool enabled = [] { const char* v = getenv("SLF_INDIRECT_SLOTS"); return !(v && atoi(v) == 0); }();
  if (!enabled || !nodes || !m->geo.indirect) return nullptr;
  if (m->slots.nodes == nodes) return &m->slots;
  if (m->geo.lat_ny > 65535 || m->geo.lat_nz > 65535) return nullptr;
  if (m->slots_mem) {
    hipFree(m->slots_mem);
    m->slots_mem = nullptr;
  }
  m->slots = slf::SlotTable{};
  const size_t n = m->geo.dist_size;
  char* mem = nullptr;
  if (hipMalloc((void**)&mem, 2 * n * sizeof(uint32_t) + 256) != hipSuccess) return nullptr;
  uint32_t* max_slot = (uint32_t*)(mem + 2 * n * sizeof(uint32_t));
  uint32_t h = 0;
  hipError_t e = hipMemsetAsync(mem, 0xFF, 2 * n * sizeof(uint32_t), s);
  if (e == hipSuccess) e = hipMemsetAsync(max_slot, 0, 256, s);
  if (e == hipSuccess) e = slf::launch_build_slot_table(m->geo, nodes, (uint32_t*)mem, (uint32_t*)mem + n, max_slot, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&h, max_slot, sizeof(h), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    hipFree(mem);
    return nullptr;
  }
  m->slots_mem = mem;
  m->slots.nodes = nodes;
  m->slots.slot_gi = (const uint32_t*)mem;
  m->slots.slot_yz = (const uint32_t*)mem + n;
  m->slots.n_slots = h + 1;
  return &m->slots;
}

extern "C" {

int slf_abi_version(void) { return SLF_ABI_VERSION; }

const char* slf_last_error(void) { return g_last_error.c_str(); }

int slf_device_count(int* count) {
  if (!count) return fail(SLF_ERR_INVALID, "count is NULL");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return hip_fail(e, "hipGetDeviceCount");
  }
  return SLF_OK;
}

int slf_device_pci_bus_id(int device, char* out, size_t len) {
  if (!out || len < 13) return fail(SLF_ERR_INVALID, "buffer of at least 13 bytes needed");
  SLF_HIP(hipDeviceGetPCIBusId(out, (int)len, device));
  return SLF_OK;
}

int slf_ctx_create(int device, slf_ctx** out) {
  if (!out) return fail(SLF_ERR_INVALID, "out is NULL");
  int n = 0;
  SLF_HIP(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(SLF_ERR_INVALID, "no such HIP device");
  SLF_HIP(hipSetDevice(device));
  slf_ctx* c = new slf_ctx;
  c->device = device;
  *out = c;
  return SLF_OK;
}

int slf_ctx_destroy(slf_ctx* ctx) {
  delete ctx;
  return SLF_OK;
}

int slf_ctx_sync(slf_ctx* ctx) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  SLF_HIP(hipSetDevice(ctx->device));
  SLF_HIP(hipDeviceSynchronize());
  return SLF_OK;
}

int slf_ctx_info(slf_ctx* ctx, char* name, size_t name_len, size_t* total_mem, int* cu_count, int* wavefront) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  hipDeviceProp_t prop;
  SLF_HIP(hipGetDeviceProperties(&prop, ctx->device));
  if (name && name_len) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (total_mem) *total_mem = prop.totalGlobalMem;
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (wavefront) *wavefront = prop.warpSize;
  return SLF_OK;
}

int slf_ctx_free_memory(slf_ctx* ctx, size_t* free_bytes) {
  if (!ctx || !free_bytes) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipSetDevice(ctx->device));
  size_t total = 0;
  SLF_HIP(hipMemGetInfo(free_bytes, &total));
  return SLF_OK;
}

int slf_malloc(slf_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipSetDevice(ctx->device));
  SLF_HIP(hipMalloc(dptr, bytes ? bytes : 1));
  return SLF_OK;
}

int slf_free(slf_ctx* ctx, void* dptr) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  if (dptr) SLF_HIP(hipFree(dptr));
  return SLF_OK;
}

int slf_memset(slf_ctx* ctx, void* dptr, int value, size_t bytes, slf_stream* stream) {
  if (!ctx || !dptr) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipMemsetAsync(dptr, value, bytes, native(stream)));
  if (!stream) SLF_HIP(hipStreamSynchronize(nullptr));   // no stream given: complete before returning
  return SLF_OK;
}

// ---- placed allocations: one virtual range, backed by separately created physical chunks -----------------
// (HIP virtual memory management: hipMemAddressReserve / hipMemCreate / hipMemMap)
int slf_vmm_granularity(slf_ctx* ctx, size_t* bytes) {
  if (!ctx || !bytes) return fail(SLF_ERR_INVALID, "NULL argument");
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = ctx->device;
  SLF_HIP(hipMemGetAllocationGranularity(bytes, &prop, hipMemAllocationGranularityRecommended));
  return SLF_OK;
}

int slf_vmm_reserve(slf_ctx* ctx, size_t bytes, void** va) {
  if (!ctx || !va) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipSetDevice(ctx->device));
  SLF_HIP(hipMemAddressReserve(va, bytes, 0, nullptr, 0));
  return SLF_OK;
}

int slf_vmm_release_range(slf_ctx* ctx, void* va, size_t bytes) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  SLF_HIP(hipMemAddressFree(va, bytes));
  return SLF_OK;
}

int slf_vmm_chunk_create(slf_ctx* ctx, size_t bytes, uint64_t* handle) {
  if (!ctx || !handle) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipSetDevice(ctx->device));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = ctx->device;
  hipMemGenericAllocationHandle_t h;
  SLF_HIP(hipMemCreate(&h, bytes, &prop, 0));
  static_assert(sizeof(h) <= sizeof(uint64_t), "allocation handle does not fit the ABI type");
  *handle = 0;
  memcpy(handle, &h, sizeof(h));
  return SLF_OK;
}

int slf_vmm_chunk_release(slf_ctx* ctx, uint64_t handle) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  hipMemGenericAllocationHandle_t h;
  memcpy(&h, &handle, sizeof(h));
  SLF_HIP(hipMemRelease(h));
  return SLF_OK;
}

int slf_vmm_map(slf_ctx* ctx, void* va, size_t bytes, uint64_t handle) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  SLF_HIP(hipSetDevice(ctx->device));
  hipMemGenericAllocationHandle_t h;
  memcpy(&h, &handle, sizeof(h));
  SLF_HIP(hipMemMap(va, bytes, 0, h, 0));
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = ctx->device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  SLF_HIP(hipMemSetAccess(va, bytes, &acc, 1));
  return SLF_OK;
}

int slf_vmm_unmap(slf_ctx* ctx, void* va, size_t bytes) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  SLF_HIP(hipSetDevice(ctx->device));
  SLF_HIP(hipMemUnmap(va, bytes));
  return SLF_OK;
}

// ---- device-to-device halo exchange over RCCL (xGMI) --------------------------------------------------------
// RCCL is bound at run time (dlopen): the library has no link-time dependency on it, and a process that already
// carries an RCCL (PyTorch bundles one) keeps exactly that one.
namespace {
struct RcclUniqueId {      // ncclUniqueId: 128 opaque bytes, passed by value
  char b[128];
};
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
  if (g_rccl.handle) return SLF_OK;
  // exactly one RCCL per process, and the one that belongs to the HIP runtime in use: the copy next to the
  // libamdhip64 this library is bound to (a PyTorch wheel bundles both; torch.distributed will load that same
  // file), then whatever the loader finds
  std::vector<std::string> names;
  Dl_info info;
  if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) names.push_back(dir.substr(0, slash) + "/librccl.so");
  }
  names.push_back("librccl.so");
  names.push_back("librccl.so.1");
  void* h = nullptr;
  for (const std::string& n : names) {           // one that is loaded already wins
    h = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (size_t i = 0; !h && i < names.size(); i++) h = dlopen(names[i].c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(SLF_ERR_NOT_FOUND, std::string("librccl not found: ") + dlerror());
#define SLF_SYM(field, name)                                                        \
  *(void**)(&g_rccl.field) = dlsym(h, name);                                        \
  if (!g_rccl.field) return fail(SLF_ERR_NOT_FOUND, std::string("librccl lacks ") + name);
  SLF_SYM(GetUniqueId, "ncclGetUniqueId")
  SLF_SYM(CommInitRank, "ncclCommInitRank")
  SLF_SYM(CommDestroy, "ncclCommDestroy")
  SLF_SYM(CommCount, "ncclCommCount")
  SLF_SYM(CommUserRank, "ncclCommUserRank")
  SLF_SYM(Send, "ncclSend")
  SLF_SYM(Recv, "ncclRecv")
  SLF_SYM(GroupStart, "ncclGroupStart")
  SLF_SYM(GroupEnd, "ncclGroupEnd")
  SLF_SYM(GetErrorString, "ncclGetErrorString")
#undef SLF_SYM
  g_rccl.handle = h;
  return SLF_OK;
}

int rccl_fail(int rc, const char* what) {
  return fail(SLF_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
}
}  // namespace

struct slf_comm {
  slf_ctx* ctx;
  void* comm;
  int nranks, rank;
};

int slf_comm_unique_id(void* id128) {
  if (!id128) return fail(SLF_ERR_INVALID, "id buffer is NULL");
  if (int e = rccl_load()) return e;
  int rc = g_rccl.GetUniqueId(id128);
  return rc ? rccl_fail(rc, "ncclGetUniqueId") : SLF_OK;
}

int slf_comm_init(slf_ctx* ctx, int nranks, int rank, const void* unique_id, slf_comm** out) {
  if (!ctx || !unique_id || !out) return fail(SLF_ERR_INVALID, "NULL argument");
  if (int e = rccl_load()) return e;
  SLF_HIP(hipSetDevice(ctx->device));
  RcclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  void* comm = nullptr;
  int rc = g_rccl.CommInitRank(&comm, nranks, id, rank);
  if (rc) return rccl_fail(rc, "ncclCommInitRank");
  *out = new slf_comm{ctx, comm, nranks, rank};
  return SLF_OK;
}

int slf_comm_destroy(slf_comm* c) {
  if (!c) return SLF_OK;
  int rc = g_rccl.CommDestroy(c->comm);
  delete c;
  return rc ? rccl_fail(rc, "ncclCommDestroy") : SLF_OK;
}

// What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank), not what the caller passed to
// slf_comm_init(): the evidence a benchmark line carries that RCCL saw N ranks.
int slf_comm_count(slf_comm* c, int* nranks, int* rank) {
  if (!c) return fail(SLF_ERR_INVALID, "comm is NULL");
  int n = 0, r = 0;
  int rc = g_rccl.CommCount(c->comm, &n);
  if (rc) return rccl_fail(rc, "ncclCommCount");
  rc = g_rccl.CommUserRank(c->comm, &r);
  if (rc) return rccl_fail(rc, "ncclCommUserRank");
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  return SLF_OK;
}

int slf_comm_group_begin(void) {
  if (int e = rccl_load()) return e;
  int rc = g_rccl.GroupStart();
  return rc ? rccl_fail(rc, "ncclGroupStart") : SLF_OK;
}

int slf_comm_group_end(void) {
  if (int e = rccl_load()) return e;
  int rc = g_rccl.GroupEnd();
  return rc ? rccl_fail(rc, "ncclGroupEnd") : SLF_OK;
}

int slf_comm_sendrecv(slf_comm* c, int peer, const void* send_dptr, size_t n_send, void* recv_dptr, size_t n_recv,
                      int elem_bytes, slf_stream* stream) {
  if (!c) return fail(SLF_ERR_INVALID, "comm is NULL");
  if (elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 1) return fail(SLF_ERR_INVALID, "elem_bytes must be 1, 4 or 8");
  const int dtype = elem_bytes == 4 ? 7 /* ncclFloat32 */ : (elem_bytes == 8 ? 8 /* ncclFloat64 */ : 0 /* ncclInt8 */);
  SLF_HIP(hipSetDevice(c->ctx->device));
  int rc = g_rccl.GroupStart();
  if (rc) return rccl_fail(rc, "ncclGroupStart");
  if (n_send) rc = g_rccl.Send(send_dptr, n_send, dtype, peer, c->comm, native(stream));
  if (!rc && n_recv) rc = g_rccl.Recv(recv_dptr, n_recv, dtype, peer, c->comm, native(stream));
  int rc2 = g_rccl.GroupEnd();
  if (rc) return rccl_fail(rc, "ncclSend/ncclRecv");
  return rc2 ? rccl_fail(rc2, "ncclGroupEnd") : SLF_OK;
}

// One RCCL group of n point-to-point operations, posted in the given order (= the matching order between a pair of
// ranks) on `stream`: the whole batch of a halo exchange in one call.
int slf_comm_exchange(slf_comm* c, const slf_comm_op* ops, int n, slf_stream* stream) {
  if (!c || (!ops && n > 0)) return fail(SLF_ERR_INVALID, "NULL argument");
  if (n <= 0) return SLF_OK;
  SLF_HIP(hipSetDevice(c->ctx->device));
  int rc = g_rccl.GroupStart();
  if (rc) return rccl_fail(rc, "ncclGroupStart");
  for (int i = 0; i < n && !rc; i++) {
    const slf_comm_op& o = ops[i];
    if (o.elem_bytes != 4 && o.elem_bytes != 8 && o.elem_bytes != 1) {
      g_rccl.GroupEnd();
      return fail(SLF_ERR_INVALID, "elem_bytes must be 1, 4 or 8");
    }
    const int dtype = o.elem_bytes == 4 ? 7 /* ncclFloat32 */ : (o.elem_bytes == 8 ? 8 /* ncclFloat64 */ : 0 /* ncclInt8 */);
    if (o.count == 0) continue;
    if (o.kind == SLF_COMM_SEND) rc = g_rccl.Send(o.dptr, o.count, dtype, o.peer, c->comm, native(stream));
    else rc = g_rccl.Recv(o.dptr, o.count, dtype, o.peer, c->comm, native(stream));
  }
  const int rc2 = g_rccl.GroupEnd();
  if (rc) return rccl_fail(rc, "ncclSend/ncclRecv");
  return rc2 ? rccl_fail(rc2, "ncclGroupEnd") : SLF_OK;
}

// ---- peer transport: receive buffers mapped into the neighbouring processes, progress counters instead of messages ---
// (include/sailfish_hip.h "peer transport".)  The producers of a halo store into the neighbour's memory themselves;
// what these entry points add is the ordering.
namespace {
typedef unsigned long long peer_u64;
constexpr int PEER_SLOT = 8;        // 64-bit words per counter: every counter has a 64-byte line to itself
constexpr int PEER_MAX = 32;        // ranks per signal / wait call (a block decomposition has at most 26 neighbours)

struct PeerSignalArgs {
  peer_u64* flag[PEER_MAX];
  peer_u64 value[PEER_MAX];
  int n;
};
struct PeerWaitArgs {
  const peer_u64* flag[PEER_MAX];
  peer_u64 value[PEER_MAX];
  int rank[PEER_MAX];
  int n, channel;
  peer_u64 timeout;                 // ticks of the 100 MHz wall clock
  volatile long long* status;       // pinned host memory
};

// One lane per counter.  The release makes everything this PROCESS enqueued before the launch on the same stream
// visible first (slf_peer_signal records an event in front: a system-scope release of the whole device, not only of
// the one XCD this wave runs on).
__global__ void peer_signal_kernel(PeerSignalArgs a) {
  const int i = threadIdx.x;
  if (i < a.n) __hip_atomic_store(a.flag[i], a.value[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One lane per counter, spinning (with sleeps) until it has reached its value -- bounded: after `timeout` ticks the lane
// reports and gives up, so that a neighbour that died leaves an error on the host instead of a device nobody can use.
__global__ void peer_wait_kernel(PeerWaitArgs a) {
  const int i = threadIdx.x;
  if (i < a.n) {
    const peer_u64 t0 = wall_clock64();
    peer_u64 seen;
    while ((seen = __hip_atomic_load(a.flag[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) < a.value[i]) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > a.timeout) {
        a.status[1] = a.rank[i];
        a.status[2] = a.channel;
        a.status[3] = (long long)a.value[i];
        a.status[4] = (long long)seen;
        a.status[5] = a.status[5] + 1;
        a.status[0] = 1;
        break;
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

__global__ void peer_fill_kernel(uint32_t* dst, size_t n, uint32_t pattern, int one_word_per_block) {
  if (one_word_per_block) {
    const size_t i = blockIdx.x;
    if (threadIdx.x == 0 && i < n) dst[i] = pattern ^ (uint32_t)i;
    return;
  }
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = pattern ^ (uint32_t)i;
}

__global__ void peer_check_kernel(const uint32_t* src, size_t n, uint32_t pattern, uint32_t* bad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && src[i] != (pattern ^ (uint32_t)i)) atomicAdd(bad, 1u);
}
}  // namespace

struct slf_peer {
  slf_ctx* ctx;
  int nranks, rank;
  peer_u64* flags;                       // my block: counter[src rank][channel], PEER_SLOT words each
  std::vector<peer_u64*> remote;         // the blocks of the other ranks as mapped here (mine: flags)
  std::vector<peer_u64> sent, awaited;   // [rank * SLF_PEER_CHANNELS + channel]
  long long* status;                     // pinned host memory, 8 words
  peer_u64 timeout;
  hipEvent_t ev;
  uint32_t* bad;                         // device word of the self-test
};

static int peer_check_ranks(slf_peer* p, const int32_t* ranks, int n, int channel) {
  if (!p || (!ranks && n > 0)) return fail(SLF_ERR_INVALID, "NULL argument");
  if (n < 0 || n > PEER_MAX) return fail(SLF_ERR_INVALID, "at most 32 ranks per peer signal / wait");
  if (channel < 0 || channel >= SLF_PEER_CHANNELS) return fail(SLF_ERR_INVALID, "bad peer channel");
  for (int i = 0; i < n; i++) {
    if (ranks[i] < 0 || ranks[i] >= p->nranks) return fail(SLF_ERR_INVALID, "peer rank out of range");
    if (!p->remote[ranks[i]]) return fail(SLF_ERR_INVALID, "peer rank not connected (slf_peer_connect)");
  }
  return SLF_OK;
}

int slf_peer_create(slf_ctx* ctx, int nranks, int rank, slf_peer** out) {
  if (!ctx || !out) return fail(SLF_ERR_INVALID, "NULL argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(SLF_ERR_INVALID, "bad rank / nranks");
  SLF_HIP(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)nranks * SLF_PEER_CHANNELS * PEER_SLOT * sizeof(peer_u64);
  void* f = nullptr;
  // uncached: a counter stored by another process (another device) is seen by the spinning lane without any cache
  // maintenance; fine-grained as the second choice
  if (hipExtMallocWithFlags(&f, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    SLF_HIP(hipExtMallocWithFlags(&f, bytes, hipDeviceMallocFinegrained));
  }
  SLF_HIP(hipMemset(f, 0, bytes));
  slf_peer* p = new slf_peer;
  p->ctx = ctx;
  p->nranks = nranks;
  p->rank = rank;
  p->flags = (peer_u64*)f;
  p->remote.assign(nranks, nullptr);
  p->remote[rank] = p->flags;
  p->sent.assign((size_t)nranks * SLF_PEER_CHANNELS, 0);
  p->awaited.assign((size_t)nranks * SLF_PEER_CHANNELS, 0);
  p->status = nullptr;
  p->timeout = 60ull * 100000000ull;
  p->ev = nullptr;
  p->bad = nullptr;
  hipError_t e = hipHostMalloc((void**)&p->status, 8 * sizeof(long long), hipHostMallocMapped);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&p->bad, sizeof(uint32_t));
  if (e != hipSuccess) {
    slf_peer_destroy(p);
    return hip_fail(e, "slf_peer_create");
  }
  memset(p->status, 0, 8 * sizeof(long long));
  SLF_HIP(hipDeviceSynchronize());
  *out = p;
  return SLF_OK;
}

int slf_peer_destroy(slf_peer* p) {
  if (!p) return SLF_OK;
  for (int r = 0; r < p->nranks; r++)
    if (r != p->rank && p->remote[r]) hipIpcCloseMemHandle(p->remote[r]);
  if (p->flags) hipFree(p->flags);
  if (p->status) hipHostFree(p->status);
  if (p->ev) hipEventDestroy(p->ev);
  if (p->bad) hipFree(p->bad);
  delete p;
  return SLF_OK;
}

int slf_peer_flags_handle(slf_peer* p, void* handle64) {
  if (!p || !handle64) return fail(SLF_ERR_INVALID, "NULL argument");
  static_assert(sizeof(hipIpcMemHandle_t) == SLF_PEER_HANDLE_BYTES, "IPC handle size");
  hipIpcMemHandle_t h;
  SLF_HIP(hipIpcGetMemHandle(&h, p->flags));
  memcpy(handle64, &h, sizeof(h));
  return SLF_OK;
}

int slf_peer_connect(slf_peer* p, int rank, const void* handle64) {
  if (!p || !handle64) return fail(SLF_ERR_INVALID, "NULL argument");
  if (rank < 0 || rank >= p->nranks) return fail(SLF_ERR_INVALID, "peer rank out of range");
  if (rank == p->rank || p->remote[rank]) return SLF_OK;
  SLF_HIP(hipSetDevice(p->ctx->device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  void* m = nullptr;
  SLF_HIP(hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess));
  p->remote[rank] = (peer_u64*)m;
  return SLF_OK;
}

int slf_peer_alloc(slf_peer* p, size_t bytes, void** dptr, void* handle64) {
  if (!p || !dptr || !handle64) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipSetDevice(p->ctx->device));
  void* m = nullptr;
  SLF_HIP(hipMalloc(&m, bytes ? bytes : 1));      // plain (cacheable) memory: one-word stores merge into lines in the L2
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, m);
  if (e != hipSuccess) {
    hipFree(m);
    return hip_fail(e, "hipIpcGetMemHandle");
  }
  memcpy(handle64, &h, sizeof(h));
  *dptr = m;
  return SLF_OK;
}

int slf_peer_free(slf_peer* p, void* dptr) {
  if (!p) return fail(SLF_ERR_INVALID, "peer is NULL");
  if (dptr) SLF_HIP(hipFree(dptr));
  return SLF_OK;
}

int slf_peer_open(slf_peer* p, const void* handle64, void** mapped) {
  if (!p || !handle64 || !mapped) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipSetDevice(p->ctx->device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  SLF_HIP(hipIpcOpenMemHandle(mapped, h, hipIpcMemLazyEnablePeerAccess));
  return SLF_OK;
}

int slf_peer_close(slf_peer* p, void* mapped) {
  if (!p) return fail(SLF_ERR_INVALID, "peer is NULL");
  if (mapped) SLF_HIP(hipIpcCloseMemHandle(mapped));
  return SLF_OK;
}

int slf_peer_signal(slf_peer* p, const int32_t* ranks, int n, int channel, slf_stream* stream) {
  if (int e = peer_check_ranks(p, ranks, n, channel)) return e;
  if (n == 0) return SLF_OK;
  SLF_HIP(hipSetDevice(p->ctx->device));
  // the stream's work so far, released to system scope by the command processor (every XCD's L2, not only the one the
  // signalling wave will run on): the writes into the neighbours' buffers are in memory before the counters move
  SLF_HIP(hipEventRecord(p->ev, native(stream)));
  PeerSignalArgs a;
  a.n = n;
  for (int i = 0; i < n; i++) {
    const size_t k = (size_t)ranks[i] * SLF_PEER_CHANNELS + channel;
    a.flag[i] = p->remote[ranks[i]] + ((size_t)p->rank * SLF_PEER_CHANNELS + channel) * PEER_SLOT;
    a.value[i] = ++p->sent[k];
  }
  hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(64), 0, native(stream), a);
  SLF_HIP(hipGetLastError());
  return SLF_OK;
}

int slf_peer_wait(slf_peer* p, const int32_t* ranks, int n, int channel, int count, slf_stream* stream) {
  if (int e = peer_check_ranks(p, ranks, n, channel)) return e;
  if (count < 1) return fail(SLF_ERR_INVALID, "peer wait: count must be at least 1");
  if (n == 0) return SLF_OK;
  SLF_HIP(hipSetDevice(p->ctx->device));
  PeerWaitArgs a;
  a.n = n;
  a.channel = channel;
  a.timeout = p->timeout;
  a.status = p->status;
  for (int i = 0; i < n; i++) {
    const size_t k = (size_t)ranks[i] * SLF_PEER_CHANNELS + channel;
    a.flag[i] = p->flags + k * PEER_SLOT;
    a.value[i] = (p->awaited[k] += (peer_u64)count);
    a.rank[i] = ranks[i];
  }
  hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, native(stream), a);
  SLF_HIP(hipGetLastError());
  return SLF_OK;
}

int slf_peer_set_timeout(slf_peer* p, double seconds) {
  if (!p || !(seconds > 0)) return fail(SLF_ERR_INVALID, "bad argument");
  p->timeout = (peer_u64)(seconds * 1e8);
  return SLF_OK;
}

int slf_peer_status(slf_peer* p, int64_t out[8]) {
  if (!p || !out) return fail(SLF_ERR_INVALID, "NULL argument");
  for (int i = 0; i < 8; i++) out[i] = (int64_t)((volatile long long*)p->status)[i];
  return SLF_OK;
}

int slf_peer_progress(slf_peer* p, int rank, int channel, uint64_t* sent, uint64_t* awaited, uint64_t* arrived) {
  if (!p) return fail(SLF_ERR_INVALID, "peer is NULL");
  if (rank < 0 || rank >= p->nranks || channel < 0 || channel >= SLF_PEER_CHANNELS) return fail(SLF_ERR_INVALID, "bad rank / channel");
  const size_t k = (size_t)rank * SLF_PEER_CHANNELS + channel;
  if (sent) *sent = p->sent[k];
  if (awaited) *awaited = p->awaited[k];
  if (arrived) {
    peer_u64 v = 0;
    SLF_HIP(hipMemcpy(&v, p->flags + k * PEER_SLOT, sizeof(v), hipMemcpyDeviceToHost));
    *arrived = v;
  }
  return SLF_OK;
}

int slf_peer_selftest_fill(slf_peer* p, void* dst, size_t nwords, uint32_t pattern, int one_word_per_block, slf_stream* stream) {
  if (!p || !dst) return fail(SLF_ERR_INVALID, "NULL argument");
  if (nwords == 0) return SLF_OK;
  SLF_HIP(hipSetDevice(p->ctx->device));
  const unsigned blocks = one_word_per_block ? (unsigned)nwords : (unsigned)((nwords + 255) / 256);
  hipLaunchKernelGGL(peer_fill_kernel, dim3(blocks), dim3(one_word_per_block ? 64 : 256), 0, native(stream), (uint32_t*)dst, nwords,
                     pattern, one_word_per_block);
  SLF_HIP(hipGetLastError());
  return SLF_OK;
}

int slf_peer_selftest_check(slf_peer* p, const void* src, size_t nwords, uint32_t pattern, slf_stream* stream, uint32_t* bad_words) {
  if (!p || !src || !bad_words) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipSetDevice(p->ctx->device));
  SLF_HIP(hipMemsetAsync(p->bad, 0, sizeof(uint32_t), native(stream)));
  if (nwords)
    hipLaunchKernelGGL(peer_check_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, native(stream), (const uint32_t*)src,
                       nwords, pattern, p->bad);
  SLF_HIP(hipGetLastError());
  SLF_HIP(hipMemcpyAsync(bad_words, p->bad, sizeof(uint32_t), hipMemcpyDeviceToHost, native(stream)));
  SLF_HIP(hipStreamSynchronize(native(stream)));
  return SLF_OK;
}

int slf_host_alloc_pinned(size_t bytes, void** hptr) {
  if (!hptr) return fail(SLF_ERR_INVALID, "hptr is NULL");
  SLF_HIP(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
  return SLF_OK;
}

int slf_host_free(void* hptr) {
  if (hptr) SLF_HIP(hipHostFree(hptr));
  return SLF_OK;
}

int slf_memcpy_h2d(slf_ctx* ctx, void* dptr, const void* hptr, size_t bytes) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  SLF_HIP(hipMemcpy(dptr, hptr, bytes, hipMemcpyHostToDevice));
  return SLF_OK;
}

int slf_memcpy_d2h(slf_ctx* ctx, void* hptr, const void* dptr, size_t bytes) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  SLF_HIP(hipMemcpy(hptr, dptr, bytes, hipMemcpyDeviceToHost));
  return SLF_OK;
}

int slf_memcpy_h2d_async(slf_ctx* ctx, void* dptr, const void* hptr, size_t bytes, slf_stream* s) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  SLF_HIP(hipSetDevice(ctx->device));
  SLF_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, native(s)));
  return SLF_OK;
}

int slf_memcpy_d2h_async(slf_ctx* ctx, void* hptr, const void* dptr, size_t bytes, slf_stream* s) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  SLF_HIP(hipSetDevice(ctx->device));
  SLF_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, native(s)));
  return SLF_OK;
}

int slf_memcpy_d2d_async(slf_ctx* ctx, void* dst, const void* src, size_t bytes, slf_stream* s) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  SLF_HIP(hipSetDevice(ctx->device));
  SLF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, native(s)));
  return SLF_OK;
}

int slf_memcpy_peer_async(slf_ctx* ctx, void* dst, int dst_device, const void* src, int src_device, size_t bytes,
                          slf_stream* s) {
  if (!ctx) return fail(SLF_ERR_INVALID, "ctx is NULL");
  SLF_HIP(hipSetDevice(ctx->device));
  SLF_HIP(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, native(s)));
  return SLF_OK;
}

static int stream_create(slf_ctx* ctx, bool high_priority, slf_stream** out) {
  if (!ctx || !out) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipSetDevice(ctx->device));
  int least = 0, greatest = 0;
  if (high_priority) SLF_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  slf_stream* s = new slf_stream;
  s->ctx = ctx;
  hipError_t e = high_priority ? hipStreamCreateWithPriority(&s->s, hipStreamNonBlocking, greatest)
                               : hipStreamCreateWithFlags(&s->s, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete s;
    return hip_fail(e, "hipStreamCreate");
  }
  *out = s;
  return SLF_OK;
}

int slf_stream_create(slf_ctx* ctx, slf_stream** out) { return stream_create(ctx, false, out); }

// The halo stream: its small pack / unpack kernels (and whatever waits behind them) are dispatched ahead of the bulk
// sweep's remaining workgroups instead of queueing behind a full chip.
int slf_stream_create_high_priority(slf_ctx* ctx, slf_stream** out) { return stream_create(ctx, true, out); }

int slf_stream_destroy(slf_stream* s) {
  if (s) {
    hipStreamDestroy(s->s);
    delete s;
  }
  return SLF_OK;
}

int slf_stream_sync(slf_stream* s) {
  SLF_HIP(hipStreamSynchronize(native(s)));
  return SLF_OK;
}

int slf_stream_native(slf_stream* s, void** hip_stream) {
  if (!hip_stream) return fail(SLF_ERR_INVALID, "hip_stream is NULL");
  *hip_stream = (void*)native(s);
  return SLF_OK;
}

int slf_graph_capture_begin(slf_stream* s) {
  if (!s) return fail(SLF_ERR_INVALID, "graph capture needs a stream created with slf_stream_create");
  // relaxed: a destructor that frees device memory while we record (Python's garbage collector can run one at
  // any time) must not invalidate the capture
  SLF_HIP(hipStreamBeginCapture(s->s, hipStreamCaptureModeRelaxed));
  return SLF_OK;
}

int slf_graph_capture_end(slf_stream* s, slf_graph** out) {
  if (!s || !out) return fail(SLF_ERR_INVALID, "NULL argument");
  hipGraph_t graph = nullptr;
  SLF_HIP(hipStreamEndCapture(s->s, &graph));
  if (!graph) return fail(SLF_ERR_HIP, "stream capture produced no graph");
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    hipGraphDestroy(graph);
    return fail(SLF_ERR_HIP, hipGetErrorString(e));
  }
  slf_graph* g = new slf_graph;
  g->graph = graph;
  g->exec = exec;
  *out = g;
  return SLF_OK;
}

int slf_graph_launch(slf_graph* g, slf_stream* s) {
  if (!g) return fail(SLF_ERR_INVALID, "graph is NULL");
  if (s && s->ctx) SLF_HIP(hipSetDevice(s->ctx->device));
  SLF_HIP(hipGraphLaunch(g->exec, native(s)));
  return SLF_OK;
}

int slf_graph_destroy(slf_graph* g) {
  if (!g) return SLF_OK;
  hipGraphExecDestroy(g->exec);
  hipGraphDestroy(g->graph);
  delete g;
  return SLF_OK;
}

int slf_stream_wait_event(slf_stream* s, slf_event* ev) {
  if (!ev) return fail(SLF_ERR_INVALID, "event is NULL");
  if (s && s->ctx) SLF_HIP(hipSetDevice(s->ctx->device));      // the waiting stream's device; the event may belong to another
  SLF_HIP(hipStreamWaitEvent(native(s), ev->e, 0));
  return SLF_OK;
}

int slf_event_create(slf_ctx* ctx, int timing, slf_event** out) {
  if (!ctx || !out) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipSetDevice(ctx->device));
  slf_event* ev = new slf_event;
  ev->ctx = ctx;
  hipError_t e = hipEventCreateWithFlags(&ev->e, timing ? hipEventDefault : hipEventDisableTiming);
  if (e != hipSuccess) {
    delete ev;
    return hip_fail(e, "hipEventCreateWithFlags");
  }
  *out = ev;
  return SLF_OK;
}

int slf_event_destroy(slf_event* ev) {
  if (ev) {
    hipEventDestroy(ev->e);
    delete ev;
  }
  return SLF_OK;
}

int slf_event_record(slf_event* ev, slf_stream* s) {
  if (!ev) return fail(SLF_ERR_INVALID, "event is NULL");
  if (ev->ctx) SLF_HIP(hipSetDevice(ev->ctx->device));
  SLF_HIP(hipEventRecord(ev->e, native(s)));
  return SLF_OK;
}

int slf_event_sync(slf_event* ev) {
  if (!ev) return fail(SLF_ERR_INVALID, "event is NULL");
  SLF_HIP(hipEventSynchronize(ev->e));
  return SLF_OK;
}

int slf_event_elapsed_ms(slf_event* start, slf_event* end, float* ms) {
  if (!start || !end || !ms) return fail(SLF_ERR_INVALID, "NULL argument");
  SLF_HIP(hipEventElapsedTime(ms, start->e, end->e));
  return SLF_OK;
}

int slf_module_create(slf_ctx* ctx, const slf_module_desc* d, slf_module** out) {
  if (!ctx || !d || !out) return fail(SLF_ERR_INVALID, "NULL argument");
  if (d->struct_size != sizeof(slf_module_desc)) return fail(SLF_ERR_INVALID, "slf_module_desc size mismatch (ABI)");
  if (d->lattice != SLF_D2Q9 && d->lattice != SLF_D3Q19) return fail(SLF_ERR_UNSUPPORTED, "unsupported lattice");
  if (d->model != SLF_BGK && d->model != SLF_MRT) return fail(SLF_ERR_UNSUPPORTED, "unsupported collision model");
  if (d->precision != 4 && d->precision != 8) return fail(SLF_ERR_UNSUPPORTED, "precision must be 4 or 8");
  if (d->access_pattern != SLF_AB && d->access_pattern != SLF_AA)
    return fail(SLF_ERR_UNSUPPORTED, "unsupported access pattern");
  const int dim = d->lattice == SLF_D2Q9 ? 2 : 3;
  if (d->envelope != 1) return fail(SLF_ERR_UNSUPPORTED, "envelope size must be 1");
  if (d->lat_nx < 3 || d->lat_ny < 3 || (dim == 3 && d->lat_nz < 3) || (dim == 2 && d->lat_nz != 1))
    return fail(SLF_ERR_INVALID, "bad lattice size");
  if (d->arr_nx < d->lat_nx || d->arr_ny < d->lat_ny || d->arr_nz < d->lat_nz)
    return fail(SLF_ERR_INVALID, "padded size smaller than lattice size");
  if (d->tau <= 0.5 && d->relaxation_enabled) return fail(SLF_ERR_INVALID, "tau must be > 0.5");
  if (d->n_types < 0 || d->n_types > SLF_MAX_NODE_TYPES) return fail(SLF_ERR_INVALID, "too many node types");
  const unsigned long long total = (unsigned long long)d->arr_nx * d->arr_ny * d->arr_nz;
  if (total >= 0xFFFFFFFFull || d->dist_stride >= 0xFFFFFFFFull)
    return fail(SLF_ERR_UNSUPPORTED, "subdomain too large for 32-bit node indices");

  slf_module* m = new slf_module;
  m->ctx = ctx;
  m->sel.lattice = d->lattice;
  m->sel.model = d->model;
  m->sel.precision = d->precision;
  m->sel.general = !d->fluid_only;
  if (d->node_addressing != SLF_ADDR_DIRECT && d->node_addressing != SLF_ADDR_INDIRECT) {
    delete m;
    return fail(SLF_ERR_INVALID, "node_addressing must be SLF_ADDR_DIRECT or SLF_ADDR_INDIRECT");
  }
  if (d->node_addressing == SLF_ADDR_INDIRECT) {
    if (d->fluid_only) { delete m; return fail(SLF_ERR_INVALID, "indirect addressing needs the node map (fluid_only = 0)"); }
    if (d->simtype != SLF_SIM_LBM && d->simtype != SLF_SIM_SHAN_CHEN_BINARY && d->simtype != SLF_SIM_SHAN_CHEN_SINGLE) {
      delete m;
      return fail(SLF_ERR_UNSUPPORTED, "indirect addressing: unknown simtype");
    }
  }
  m->access_pattern = d->access_pattern;
  slf::Geometry& g = m->geo;
  g.dim = dim;
  g.lat_nx = d->lat_nx; g.lat_ny = d->lat_ny; g.lat_nz = d->lat_nz;
  g.arr_nx = d->arr_nx; g.arr_ny = d->arr_ny; g.arr_nz = d->arr_nz;
  g.arr_nxy = d->arr_nx * d->arr_ny;
  g.dist_size = (uint32_t)(d->dist_stride ? d->dist_stride : total);
  if (d->node_addressing == SLF_ADDR_INDIRECT) {
    // the stride is the number of active-node slots: the caller's business (it owns the address table)
    if (d->dist_stride == 0) { delete m; return fail(SLF_ERR_INVALID, "indirect addressing: dist_stride (number of slots) must be given"); }
  } else if (g.dist_size < total) {
    delete m;
    return fail(SLF_ERR_INVALID, "dist_stride smaller than the subdomain");
  }
  for (int i = 0; i < 3; i++) {
    g.wrap[i] = (i < dim) ? (d->periodic_fused[i] != 0) : 0;
    g.axis_mode[i] = g.wrap[i] ? 2 : ((i < dim && d->periodic_local[i]) ? 1 : 0);
  }
  g.type_mask = d->nt_type_mask;
  g.param_shift = d->nt_misc_shift;
  g.param_mask = (1u << d->nt_param_shift) - 1u;
  g.orient_shift = d->nt_misc_shift + d->nt_param_shift + d->nt_scratch_shift;
  g.type_lut = 0;
  for (int i = 0; i < d->n_types; i++) {
    const int k = d->type_kind[i];
    if (k < 0 || k >= slf::NK_COUNT) {
      delete m;
      return fail(SLF_ERR_UNSUPPORTED, "node type kind not supported by the HIP backend");
    }
    g.type_lut |= (unsigned long long)k << (4 * i);
  }
  for (int i = 0; i < d->n_types; i++) {
    const int k = d->type_kind[i];
    if ((k == SLF_NK_COPY || k == SLF_NK_YU_OUTFLOW) && d->access_pattern != SLF_AB) {
      delete m;
      return fail(SLF_ERR_UNSUPPORTED, "NTCopy / NTYuOutflow nodes read neighbouring nodes of the input lattice: "
                                       "two-copy (AB) access pattern only, as in the reference (boundary.mako:626-628)");
    }
  }
  for (int i = 0; i < d->n_types; i++) {
    if (d->type_kind[i] == SLF_NK_DO_NOTHING && d->access_pattern != SLF_AB && d->node_addressing == SLF_ADDR_INDIRECT) {
      delete m;
      return fail(SLF_ERR_UNSUPPORTED, "NTDoNothing nodes of the in-place (AA) pattern keep their unknown populations in the ghost "
                                       "nodes behind them, which own no slot under indirect addressing (the reference indexes "
                                       "nodes[] unguarded there, boundary.mako:862-876 with propagation.mako:93-99)");
    }
  }
  g.bc_level = 0;
  for (int i = 0; i < d->n_types; i++) {
    const int k = d->type_kind[i];
    const bool plain = k == SLF_NK_FLUID || k == SLF_NK_GHOST || k == SLF_NK_UNUSED || k == SLF_NK_PROPAGATION_ONLY ||
                       k == SLF_NK_FULL_BB;
    const int level = (k == SLF_NK_COPY || k == SLF_NK_YU_OUTFLOW || k == SLF_NK_DO_NOTHING || k == SLF_NK_SLIP) ? 2 : (plain ? 0 : 1);
    if (level > g.bc_level) g.bc_level = level;
  }
  if (const char* ev = getenv("SLF_BC_LEVEL")) {      // tests: force the full instantiation
    if (atoi(ev) > g.bc_level) g.bc_level = atoi(ev);
  }
  g.x_ghost_unused = 0;
  g.use_link_tags = d->use_link_tags;
  g.indirect = d->node_addressing == SLF_ADDR_INDIRECT;
  g.variant = SLF_DEFAULT_VARIANT;
  if (d->sparse_geometry) g.variant |= 64;
  if (const char* ev = getenv("SLF_VARIANT")) g.variant = atoi(ev);
  slf::Physics& ph = m->phys;
  ph.tau = d->tau;
  ph.visc = d->visc;
  for (int i = 0; i < 3; i++) ph.accel[i] = d->accel[i];
  for (int i = 0; i < 27; i++) ph.mrt_rates[i] = d->mrt_rates[i];
  ph.incompressible = d->incompressible;
  ph.has_force = d->has_force;
  ph.relaxation_enabled = d->relaxation_enabled;
  ph.force_edm = d->force_implementation == SLF_FORCE_EDM;
  if (ph.force_edm && d->model != SLF_BGK) {
    delete m;
    return fail(SLF_ERR_UNSUPPORTED, "the exact difference method (force_implementation = EDM) needs the BGK collision");
  }
  ph.regularized = d->regularized != 0;
  ph.subgrid = d->subgrid;
  ph.smagorinsky_const = d->smagorinsky_const;
  if (d->subgrid != SLF_SUBGRID_NONE && d->subgrid != SLF_SUBGRID_LES_SMAGORINSKY) {
    delete m;
    return fail(SLF_ERR_INVALID, "subgrid must be one of SLF_SUBGRID_*");
  }
  if (ph.regularized || ph.subgrid) {
    // the options of the BGK relaxation preamble (relaxation_common.mako:166-237).  The reference's MRT relaxation does not
    // call the preamble (its kernels would silently ignore both flags); here that is an error, as are the combinations the
    // preamble's expressions were never written for
    if (d->model != SLF_BGK || d->simtype != SLF_SIM_LBM || d->incompressible == SLF_DENSITY_ROUNDOFF || ph.force_edm) {
      delete m;
      return fail(SLF_ERR_UNSUPPORTED, "regularized / subgrid: single-fluid BGK modules, standard or incompressible density "
                                       "model, Guo forcing or no body force");
    }
    if (ph.subgrid && !(d->smagorinsky_const > 0.0)) {
      delete m;
      return fail(SLF_ERR_INVALID, "subgrid = les-smagorinsky needs smagorinsky_const > 0");
    }
    g.variant = 0;       // per-node kernels: the tuned / whole-row ones implement the plain collision
  }
  if (d->incompressible == SLF_DENSITY_ROUNDOFF) {
    // --minimize_roundoff: "BGK-like models" in the reference (lb_base.py:72-76); here BGK, single fluid, fluid and
    // bounce-back nodes, through the per-node kernels (slf_node.h: macro_roundoff, bgk_relax_roundoff)
    bool ok = d->model == SLF_BGK && d->simtype == SLF_SIM_LBM;
    for (int i = 0; ok && i < d->n_types; i++) {
      const int k = d->type_kind[i];
      ok = k == SLF_NK_FLUID || k == SLF_NK_GHOST || k == SLF_NK_UNUSED || k == SLF_NK_PROPAGATION_ONLY ||
           k == SLF_NK_FULL_BB || k == SLF_NK_HALF_BB || k == SLF_NK_EQUILIBRIUM_DENSITY || k == SLF_NK_EQUILIBRIUM_VELOCITY;
    }
    if (!ok) {
      delete m;
      return fail(SLF_ERR_UNSUPPORTED, "minimize_roundoff: BGK single-fluid modules with fluid, bounce-back and equilibrium "
                                       "density / velocity nodes only (the reference's regularized and Zou-He expressions are "
                                       "inconsistent under the option)");
    }
    g.variant = 0;       // no tuned / whole-row kernels: they implement the standard formulation
  } else if (d->incompressible != SLF_DENSITY_COMPRESSIBLE && d->incompressible != SLF_DENSITY_INCOMPRESSIBLE) {
    delete m;
    return fail(SLF_ERR_INVALID, "incompressible must be one of SLF_DENSITY_*");
  }
  m->sc.enabled = (d->simtype == SLF_SIM_SHAN_CHEN_BINARY) ? 1 : ((d->simtype == SLF_SIM_SHAN_CHEN_SINGLE) ? 2 : 0);
  m->sc.tau_phi = d->tau_phi;
  for (int i = 0; i < 4; i++) m->sc.G[i] = d->sc_G[i];
  m->sc.potential = d->sc_potential;
  for (int i = 0; i < 3; i++) m->sc.accel1[i] = d->accel1[i];
  if (m->sc.enabled) {
    if (d->model != SLF_BGK) { delete m; return fail(SLF_ERR_UNSUPPORTED, "Shan-Chen modules use the BGK collision"); }
    if (m->sc.enabled == 1 && d->tau_phi <= 0.5) { delete m; return fail(SLF_ERR_INVALID, "tau_phi must be > 0.5"); }
    if (m->sc.enabled == 2) m->sc.tau_phi = d->tau;
    for (int i = 0; i < d->n_types; i++) {
      const int k = d->type_kind[i];
      if (!(k == SLF_NK_FLUID || k == SLF_NK_GHOST || k == SLF_NK_UNUSED || k == SLF_NK_PROPAGATION_ONLY ||
            k == SLF_NK_FULL_BB)) {
        delete m;
        return fail(SLF_ERR_UNSUPPORTED, "Shan-Chen modules support fluid and full-way bounce-back nodes only");
      }
    }
  }
  // Workgroup = an x-chunk of one row.  Whole rows up to 1024 nodes are one
  // workgroup; longer rows are cut into 256-thread chunks.
  int bx = ((d->lat_nx - 2 + 63) / 64) * 64;
  if (bx > 1024) bx = 256;
  const char* env = getenv("SLF_BLOCK_X");
  if (env && atoi(env) >= 64 && atoi(env) <= 1024 && atoi(env) % 64 == 0) bx = atoi(env);
  m->block_x = bx;
  m->node_params = nullptr;
  m->status = nullptr;
  m->xsend[0] = m->xsend[1] = m->xrecv[0] = m->xrecv[1] = nullptr;
  for (int w = 0; w < 3; w++) m->sc_send[w][0] = m->sc_send[w][1] = m->sc_recv[w][0] = m->sc_recv[w][1] = nullptr;
  m->rows = slf::RowClasses{};
  m->rows_mem = nullptr;
  m->slots = slf::SlotTable{};
  m->slots_mem = nullptr;
  m->n_node_params = d->n_node_params > 0 ? d->n_node_params : 0;
  m->precision = d->precision;
  for (int i = 0; i < 4; i++) { m->stage[i] = nullptr; m->stage_ev[i] = nullptr; m->stage_bytes[i] = 0; }
  m->stage_next = 0;
  const int np = d->n_node_params > 0 ? d->n_node_params : 1;
  hipError_t e = hipSetDevice(ctx->device);
  if (e == hipSuccess) e = hipMalloc(&m->node_params, (size_t)np * d->precision);
  if (e != hipSuccess) {
    delete m;
    return hip_fail(e, "hipMalloc(node_params)");
  }
  if (d->precision == 4) {
    std::vector<float> tmp(np, 0.0f);
    for (int i = 0; i < d->n_node_params; i++) tmp[i] = (float)d->node_params[i];
    e = hipMemcpy(m->node_params, tmp.data(), np * sizeof(float), hipMemcpyHostToDevice);
  } else {
    std::vector<double> tmp(np, 0.0);
    for (int i = 0; i < d->n_node_params; i++) tmp[i] = d->node_params[i];
    e = hipMemcpy(m->node_params, tmp.data(), np * sizeof(double), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    hipFree(m->node_params);
    delete m;
    return hip_fail(e, "hipMemcpy(node_params)");
  }
  e = hipMalloc((void**)&m->status, 4 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(m->status, 0, 4 * sizeof(uint32_t));
  if (e != hipSuccess) {
    hipFree(m->node_params);
    if (m->status) hipFree(m->status);
    delete m;
    return hip_fail(e, "hipMalloc(status)");
  }
  *out = m;
  return SLF_OK;
}

int slf_module_destroy(slf_module* m) {
  if (m) {
    if (m->node_params) hipFree(m->node_params);
    if (m->status) hipFree(m->status);
    if (m->rows_mem) hipFree(m->rows_mem);
    if (m->slots_mem) hipFree(m->slots_mem);
    for (int i = 0; i < 4; i++) {
      if (m->stage[i]) hipHostFree(m->stage[i]);
      if (m->stage_ev[i]) hipEventDestroy(m->stage_ev[i]);
    }
    delete m;
  }
  return SLF_OK;
}

// Row classes of a node map (RowClasses, slf_kernels.h): which 64-node segments are plain fluid (their waves skip the
// map), which rows hold boundary-condition nodes (swept by a second launch of the full instantiation, so that all
// other rows run the small one).  Optional: without it every wave reads the map and the whole subdomain runs the
// instantiation for the module's node-type table.  Call again after the map's contents changed.
int slf_module_classify_rows(slf_module* m, const void* map_dptr, slf_stream* stream, int32_t out_counts[4]) {
  if (!m) return fail(SLF_ERR_INVALID, "module is NULL");
  const slf::Geometry& g = m->geo;
  if (m->rows_mem) {
    SLF_HIP(hipFree(m->rows_mem));
    m->rows_mem = nullptr;
  }
  m->rows = slf::RowClasses{};
  if (out_counts) out_counts[0] = out_counts[1] = out_counts[2] = out_counts[3] = 0;
  if (!map_dptr) return SLF_OK;                                   // forget the tables
  if (!m->sel.general || m->sel.lattice != 1 || g.indirect || !(g.variant & 8))
    return fail(SLF_ERR_UNSUPPORTED, "row classes: D3Q19 modules with a node map and direct addressing (whole-row kernels) only");
  SLF_HIP(hipSetDevice(m->ctx->device));
  const int nx = g.lat_nx - 2;
  const int nseg = (nx + 63) / 64;
  const size_t nrows_arr = (size_t)g.arr_ny * (size_t)g.arr_nz;
  const size_t seg_bytes = (nrows_arr * (size_t)nseg + 255) / 256 * 256 + 256;     // + slack: idle waves behind the last segment
  const size_t row_bytes = (nrows_arr + 255) / 256 * 256;
  const size_t list_bytes = ((size_t)(g.lat_ny - 2) * (size_t)(g.lat_nz - 2) * 4 + 255) / 256 * 256;
  char* mem = nullptr;
  SLF_HIP(hipMalloc((void**)&mem, seg_bytes + row_bytes + list_bytes + 256));
  hipStream_t s = native(stream);
  hipError_t e = hipMemsetAsync(mem, 1, seg_bytes + row_bytes, s);          // ghost rows: "mixed", never used
  if (e == hipSuccess) e = hipMemsetAsync(mem + seg_bytes + row_bytes, 0, list_bytes + 256, s);
  uint32_t* counters = (uint32_t*)(mem + seg_bytes + row_bytes + list_bytes);
  if (e == hipSuccess)
    e = slf::launch_classify_rows(g, map_dptr, (uint32_t*)mem, (uint8_t*)(mem + seg_bytes),
                                  (uint32_t*)(mem + seg_bytes + row_bytes), counters, nseg, s);
  uint32_t h[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(h, counters, sizeof(h), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    hipFree(mem);
    return hip_fail(e, "slf_module_classify_rows");
  }
  m->rows_mem = mem;
  m->rows.map = map_dptr;
  m->rows.seg_class = (const uint32_t*)mem;
  m->rows.row_class = (const uint8_t*)(mem + seg_bytes);
  m->rows.bc_rows = (const uint32_t*)(mem + seg_bytes + row_bytes);
  m->rows.nseg = nseg;
  m->rows.n_rows = (g.lat_ny - 2) * (g.dim == 3 ? g.lat_nz - 2 : 1);
  m->rows.n_bc_rows = (int)h[0];
  m->rows.n_fluid_segments = h[1];
  m->rows.n_segments = (long long)m->rows.n_rows * nseg;
  if (out_counts) {
    out_counts[0] = m->rows.n_rows;
    out_counts[1] = m->rows.n_bc_rows;
    out_counts[2] = (int32_t)(m->rows.n_segments > 0x7fffffff ? 0x7fffffff : m->rows.n_segments);
    out_counts[3] = (int32_t)(m->rows.n_fluid_segments > 0x7fffffff ? 0x7fffffff : m->rows.n_fluid_segments);
  }
  return SLF_OK;
}

namespace {
constexpr int PARAM_UPDATE_MAX = 256;
struct ParamUpdateF {
  float v[PARAM_UPDATE_MAX];
  int n;
};
struct ParamUpdateD {
  double v[PARAM_UPDATE_MAX];
  int n;
};
__global__ void update_params_kernel_f(float* table, ParamUpdateF u) {
  for (int i = threadIdx.x; i < u.n; i += blockDim.x) table[i] = u.v[i];
}
__global__ void update_params_kernel_d(double* table, ParamUpdateD u) {
  for (int i = threadIdx.x; i < u.n; i += blockDim.x) table[i] = u.v[i];
}
}  // namespace

int slf_module_update_node_params(slf_module* m, int first, const double* values, int n, slf_stream* stream) {
  if (!m || (!values && n > 0)) return fail(SLF_ERR_INVALID, "NULL argument");
  if (n <= 0) return SLF_OK;
  if (first < 0 || first + n > m->n_node_params) return fail(SLF_ERR_INVALID, "node parameter range outside the module's table");
  SLF_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = native(stream);
  char* dst = (char*)m->node_params + (size_t)first * m->precision;
  if (n <= PARAM_UPDATE_MAX) {
    // the values travel as kernel arguments: nothing the host must keep alive, nothing that could make the call wait
    if (m->precision == 4) {
      ParamUpdateF u;
      u.n = n;
      for (int i = 0; i < n; i++) u.v[i] = (float)values[i];
      hipLaunchKernelGGL(update_params_kernel_f, dim3(1), dim3(64), 0, s, (float*)dst, u);
    } else {
      ParamUpdateD u;
      u.n = n;
      for (int i = 0; i < n; i++) u.v[i] = values[i];
      hipLaunchKernelGGL(update_params_kernel_d, dim3(1), dim3(64), 0, s, (double*)dst, u);
    }
    SLF_HIP(hipGetLastError());
    return SLF_OK;
  }
  // long updates (a value per node of an inlet): through one of four pinned staging buffers, reused once its last copy is done
  const int k = m->stage_next;
  m->stage_next = (k + 1) & 3;
  const size_t bytes = (size_t)n * m->precision;
  if (m->stage_ev[k]) SLF_HIP(hipEventSynchronize(m->stage_ev[k]));
  else SLF_HIP(hipEventCreateWithFlags(&m->stage_ev[k], hipEventDisableTiming));
  if (m->stage_bytes[k] < bytes) {
    if (m->stage[k]) SLF_HIP(hipHostFree(m->stage[k]));
    m->stage[k] = nullptr;
    SLF_HIP(hipHostMalloc(&m->stage[k], bytes, hipHostMallocDefault));
    m->stage_bytes[k] = bytes;
  }
  if (m->precision == 4) for (int i = 0; i < n; i++) ((float*)m->stage[k])[i] = (float)values[i];
  else memcpy(m->stage[k], values, bytes);
  SLF_HIP(hipMemcpyAsync(dst, m->stage[k], bytes, hipMemcpyHostToDevice, s));
  SLF_HIP(hipEventRecord(m->stage_ev[k], s));
  return SLF_OK;
}

int slf_module_set_body_force(slf_module* m, int lattice, const double accel[3]) {
  if (!m || !accel) return fail(SLF_ERR_INVALID, "NULL argument");
  if (lattice == 0) {
    if (!m->phys.has_force) return fail(SLF_ERR_INVALID, "the module was created without a body force (has_force)");
    for (int i = 0; i < 3; i++) m->phys.accel[i] = accel[i];
  } else if (lattice == 1) {
    if (m->sc.enabled != 1) return fail(SLF_ERR_INVALID, "lattice 1 exists in binary models only");
    for (int i = 0; i < 3; i++) m->sc.accel1[i] = accel[i];
  } else {
    return fail(SLF_ERR_INVALID, "lattice must be 0 or 1");
  }
  return SLF_OK;
}

int slf_module_poll_invalid(slf_module* m, slf_stream* stream, int32_t out[4]) {
  if (!m || !out) return fail(SLF_ERR_INVALID, "NULL argument");
  uint32_t h[4] = {0, 0, 0, 0};
  SLF_HIP(hipSetDevice(m->ctx->device));
  SLF_HIP(hipMemcpyAsync(h, m->status, sizeof(h), hipMemcpyDeviceToHost, native(stream)));
  SLF_HIP(hipStreamSynchronize(native(stream)));
  for (int i = 0; i < 4; i++) out[i] = (int32_t)h[i];
  if (h[0]) SLF_HIP(hipMemsetAsync(m->status, 0, sizeof(h), native(stream)));
  return SLF_OK;
}

static int check_xface_buffers(const slf_module* m, void* send_low, void* send_high, void* recv_low, void* recv_high) {
  if (!m) return fail(SLF_ERR_INVALID, "module is NULL");
  if (send_low || send_high || recv_low || recv_high) {
    const slf::Geometry& g = m->geo;
    if (m->sel.lattice != 1 || g.indirect || m->sc.enabled || !(g.variant & 8))
      return fail(SLF_ERR_UNSUPPORTED, "x-face buffers: D3Q19 single-fluid modules with direct addressing (whole-row kernels) only");
    if (g.wrap[0]) return fail(SLF_ERR_INVALID, "x-face buffers make no sense with x wrapped inside the sweep");
    if (g.lat_nx - 2 > 1024) return fail(SLF_ERR_UNSUPPORTED, "x-face buffers: rows of at most 1024 nodes");
    if ((send_low == nullptr) != (recv_low == nullptr) || (send_high == nullptr) != (recv_high == nullptr))
      return fail(SLF_ERR_INVALID, "a connected face needs both its send and its receive buffer");
  }
  return SLF_OK;
}

int slf_module_set_xface_buffers(slf_module* m, void* send_low, void* send_high, void* recv_low, void* recv_high) {
  if (int e = check_xface_buffers(m, send_low, send_high, recv_low, recv_high)) return e;
  m->xsend[0] = send_low;
  m->xsend[1] = send_high;
  m->xrecv[0] = recv_low;
  m->xrecv[1] = recv_high;
  return SLF_OK;
}

// x-face planes of the binary model into the launch arguments; all six faces' worth or none (a sweep without the density
// planes would read ghost columns nobody fills)
static void sc_planes_of(const slf_module* m, slf::SweepArgs& a) {
  for (int f = 0; f < 2; f++) {
    a.xsend[f] = m->sc_send[0][f];   a.xrecv[f] = m->sc_recv[0][f];
    a.xsend2[f] = m->sc_send[1][f];  a.xrecv2[f] = m->sc_recv[1][f];
    a.msend[f] = m->sc_send[2][f];   a.mrecv[f] = m->sc_recv[2][f];
  }
}

static int check_xface_planes(const slf_module* m, int32_t which, void* send_low, void* send_high, void* recv_low,
                              void* recv_high) {
  if (!m) return fail(SLF_ERR_INVALID, "module is NULL");
  if (which < 0 || which > 2) return fail(SLF_ERR_INVALID, "x-face planes: 0 / 1 = populations of lattice 0 / 1, 2 = densities");
  if (send_low || send_high || recv_low || recv_high) {
    const slf::Geometry& g = m->geo;
    if (m->sel.lattice != 1 || g.indirect || !m->sc.enabled || !(g.variant & 8))
      return fail(SLF_ERR_UNSUPPORTED, "x-face planes: D3Q19 Shan-Chen modules with direct addressing (whole-row kernels)");
    if (m->sc.enabled == 2 && which == 1) return fail(SLF_ERR_INVALID, "x-face planes: the single-component model has one lattice");
    if (g.wrap[0]) return fail(SLF_ERR_INVALID, "x-face planes make no sense with x wrapped inside the sweep");
    if (g.lat_nx - 2 > 1024 || g.lat_nx - 2 < 2) return fail(SLF_ERR_UNSUPPORTED, "x-face planes: rows of 2 .. 1024 nodes");
    if ((send_low == nullptr) != (recv_low == nullptr) || (send_high == nullptr) != (recv_high == nullptr))
      return fail(SLF_ERR_INVALID, "a connected face needs both its send and its receive plane");
  }
  return SLF_OK;
}

int slf_module_set_xface_planes(slf_module* m, int32_t which, void* send_low, void* send_high, void* recv_low,
                                void* recv_high) {
  if (int e = check_xface_planes(m, which, send_low, send_high, recv_low, recv_high)) return e;
  m->sc_send[which][0] = send_low;
  m->sc_send[which][1] = send_high;
  m->sc_recv[which][0] = recv_low;
  m->sc_recv[which][1] = recv_high;
  return SLF_OK;
}

// The ghost columns x = 0 (low) / x = nx + 1 (high) of this subdomain carry nothing the simulation uses (contract:
// include/sailfish_hip.h): the whole-row sweeps neither push into them nor pull out of them.  Refused for a face the
// module itself needs: x periodic without in-sweep wrap.
int slf_module_set_x_ghost_unused(slf_module* m, int low, int high) {
  if (!m) return fail(SLF_ERR_INVALID, "module is NULL");
  if ((low || high) && m->geo.axis_mode[0] == 1)
    return fail(SLF_ERR_INVALID, "x is periodic through the ghost-layer kernels: its ghost columns are read");
  m->geo.x_ghost_unused = (low ? 1 : 0) | (high ? 2 : 0);
  return SLF_OK;
}

int slf_module_block_size(slf_module* m, int* threads) {
  if (!m || !threads) return fail(SLF_ERR_INVALID, "NULL argument");
  *threads = m->block_x;
  return SLF_OK;
}

int slf_kernel_get(slf_module* m, const char* name, slf_kernel** out) {
  if (!m || !name || !out) return fail(SLF_ERR_INVALID, "NULL argument");
  KernelKind kk;
  bool sc_local_velocity = false;
  if (!strcmp(name, "CollideAndPropagate")) kk = (m->sc.enabled == 2) ? KK_SCS_SWEEP : KK_COLLIDE_AND_PROPAGATE;
  else if (!strcmp(name, "PrepareMacroFields")) kk = KK_SCS_MACRO;
  else if (!strcmp(name, "SetInitialConditions")) kk = (m->sc.enabled == 1) ? KK_SC_INIT : KK_SET_INITIAL_CONDITIONS;
  else if (!strcmp(name, "ShanChenPrepareMacroFields")) kk = KK_SC_MACRO;
  else if (!strcmp(name, "ShanChenCollideAndPropagate0")) kk = KK_SC_SWEEP0;
  else if (!strcmp(name, "ShanChenCollideAndPropagate1")) kk = KK_SC_SWEEP1;
  else if (!strcmp(name, "ShanChenCollideAndPropagateFused")) kk = KK_SC_FUSED;
  else if (!strcmp(name, "ShanChenPrepareDensities")) { kk = KK_SC_MACRO; sc_local_velocity = true; }
  else if (!strcmp(name, "ShanChenCollideAndPropagateFusedV")) { kk = KK_SC_FUSED; sc_local_velocity = true; }
  else if (!strcmp(name, "ApplyPeriodicBoundaryConditions")) kk = KK_PBC;
  else if (!strcmp(name, "ApplyPeriodicBoundaryConditionsWithSwap")) kk = KK_PBC_SWAP;
  else if (!strcmp(name, "ApplyMacroPeriodicBoundaryConditions")) kk = KK_MACRO_PBC;
  else if (!strcmp(name, "CollectSparseData")) kk = KK_COLLECT_SPARSE;
  else if (!strcmp(name, "DistributeSparseData")) kk = KK_DISTRIBUTE_SPARSE;
  else if (!strcmp(name, "CollectContinuousData")) kk = KK_COLLECT_BOX;
  else if (!strcmp(name, "DistributeContinuousData")) kk = KK_DISTRIBUTE_BOX;
  else if (!strcmp(name, "CollectContinuousDataWithSwap")) kk = KK_COLLECT_FACE_SWAP;
  else if (!strcmp(name, "DistributeContinuousDataWithSwap")) kk = KK_DISTRIBUTE_FACE_SWAP;
  else if (!strcmp(name, "CollectContinuousMacroData")) kk = KK_COLLECT_MACRO_FACE;
  else if (!strcmp(name, "DistributeContinuousMacroData")) kk = KK_DISTRIBUTE_MACRO_FACE;
  else if (!strcmp(name, "ComputeMacroFields")) kk = KK_COMPUTE_MACRO;
  else if (!strcmp(name, "CollideAndPropagateResident")) kk = KK_RESIDENT;
  else return fail(SLF_ERR_NOT_FOUND, std::string("unknown kernel: ") + name);
  if (kk == KK_RESIDENT) {
    // what the resident kernel serves (slf_resident.hip): 2-D single-fluid modules, direct addressing, the standard
    // density formulation, node kinds whose code touches nothing but the node's own populations
    const slf::Geometry& g = m->geo;
    if (g.dim != 2 || m->sc.enabled || g.indirect || m->phys.incompressible == SLF_DENSITY_ROUNDOFF || m->phys.regularized ||
        m->phys.subgrid)
      return fail(SLF_ERR_UNSUPPORTED, "CollideAndPropagateResident: 2-D single-fluid modules with direct addressing and the "
                                       "standard density formulation only");
    if (g.axis_mode[0] == 1 || g.axis_mode[1] == 1)
      return fail(SLF_ERR_UNSUPPORTED, "CollideAndPropagateResident: periodic axes must be wrapped in-sweep (periodic_fused)");
    for (int t = 0; t < 16; t++) {
      const int kind = (int)((g.type_lut >> (4u * t)) & 0xFull);
      if (kind == slf::NK_HALF_BB || kind == slf::NK_COPY || kind == slf::NK_YU_OUTFLOW || kind == slf::NK_DO_NOTHING)
        return fail(SLF_ERR_UNSUPPORTED, "CollideAndPropagateResident: half-way bounce-back, outflow and do-nothing nodes read / "
                                         "write memory from their node code");
    }
  }
  if (kk == KK_SCS_MACRO && m->sc.enabled != 2)
    return fail(SLF_ERR_NOT_FOUND, "PrepareMacroFields only exists in single-component Shan-Chen modules");
  if ((kk == KK_SC_MACRO || kk == KK_SC_SWEEP0 || kk == KK_SC_SWEEP1 || kk == KK_SC_FUSED) && m->sc.enabled != 1)
    return fail(SLF_ERR_NOT_FOUND, "Shan-Chen kernels only exist in modules built with simtype = SLF_SIM_SHAN_CHEN_BINARY");
  if ((kk == KK_COLLIDE_AND_PROPAGATE || kk == KK_COMPUTE_MACRO) && m->sc.enabled == 1)
    return fail(SLF_ERR_NOT_FOUND, "single-fluid kernels do not exist in a Shan-Chen module");
  if ((kk == KK_PBC || kk == KK_PBC_SWAP) && m->geo.indirect)
    return fail(SLF_ERR_UNSUPPORTED, "indirect addressing: periodic boundaries are wrapped inside the sweep "
                                     "(periodic_fused), there are no ghost-layer PBC kernels");
  if (kk == KK_PBC_SWAP && m->access_pattern != SLF_AA)
    return fail(SLF_ERR_NOT_FOUND, "ApplyPeriodicBoundaryConditionsWithSwap only exists for the AA access pattern");
  if ((kk == KK_COLLECT_FACE_SWAP || kk == KK_DISTRIBUTE_FACE_SWAP) && m->access_pattern != SLF_AA)
    return fail(SLF_ERR_NOT_FOUND, "Collect / DistributeContinuousDataWithSwap only exist for the AA access pattern "
                                   "(reference kernel_utils.mako:536, 638, 700, 786)");
  if ((kk == KK_COLLECT_FACE_SWAP || kk == KK_DISTRIBUTE_FACE_SWAP || kk == KK_COLLECT_MACRO_FACE ||
       kk == KK_DISTRIBUTE_MACRO_FACE) && m->geo.indirect)
    return fail(SLF_ERR_UNSUPPORTED, "indirect addressing: halo through index lists (Collect / DistributeSparseData)");
  slf_kernel* k = new slf_kernel;
  k->mod = m;
  k->kind = kk;
  k->needs_iteration = 0;
  k->iteration = 0;
  k->bound = false;
  k->sc_local_velocity = sc_local_velocity;
  *out = k;
  return SLF_OK;
}

int slf_kernel_destroy(slf_kernel* k) {
  delete k;
  return SLF_OK;
}

int slf_kernel_set_args(slf_kernel* k, const char* fmt, const void* const* argv, int argc, int needs_iteration) {
  if (!k || !fmt || (!argv && argc > 0)) return fail(SLF_ERR_INVALID, "NULL argument");
  if ((int)strlen(fmt) != argc) return fail(SLF_ERR_INVALID, "format / argc mismatch");
  k->ptrs.clear();
  k->ints.clear();
  for (int i = 0; i < argc; i++) {
    switch (fmt[i]) {
      case 'P': k->ptrs.push_back(*(const unsigned long long*)argv[i]); break;
      case 'i': k->ints.push_back(*(const int32_t*)argv[i]); break;
      default: return fail(SLF_ERR_INVALID, std::string("unsupported format char: ") + fmt[i]);
    }
  }
  const int dim = k->mod->geo.dim;
  size_t want_p = 0, want_i = 0;
  switch (k->kind) {
    case KK_COLLIDE_AND_PROPAGATE:
    case KK_COMPUTE_MACRO: want_p = 4 + dim; want_i = 1; break;   // map, dist_in, dist_out, rho, v.., options
    case KK_SET_INITIAL_CONDITIONS: want_p = 3 + dim; want_i = 0; break;  // dist, v.., rho, map
    case KK_PBC:
    case KK_PBC_SWAP:
    case KK_MACRO_PBC: want_p = 1; want_i = 1; break;             // dist|field, axis
    case KK_COLLECT_SPARSE:
    case KK_DISTRIBUTE_SPARSE: want_p = 3; want_i = 1; break;     // idx_array, dist, buffer, n
    case KK_COLLECT_BOX:
    case KK_DISTRIBUTE_BOX:                                       // dist, buffer, dirs, base, col_stride, ncols, row_stride, nrows
      want_p = 2;                                                 // [, buffer stride between directions, between rows]
      want_i = (k->ints.size() == 8) ? 8 : 6;
      // ... or the reference's own argument list (kernel_utils.mako:526-543, 629-645, 692-708, 777-793):
      // 3-D (dist, face, base_gx, base_other, max_lx, max_other, buffer), 2-D (dist, face, base_gx, max_lx, buffer)
      if (k->ints.size() == (dim == 3 ? 5u : 3u)) {
        want_i = k->ints.size();
        if (strcmp(fmt, dim == 3 ? "PiiiiiP" : "PiiiP")) return fail(SLF_ERR_INVALID, "reference form: the buffer is the last argument");
        if (k->mod->geo.indirect) return fail(SLF_ERR_UNSUPPORTED, "indirect addressing: halo through index lists");
      }
      break;
    case KK_COLLECT_FACE_SWAP:
    case KK_DISTRIBUTE_FACE_SWAP:                                 // as the reference form above
      want_p = 2; want_i = dim == 3 ? 5 : 3;
      if (strcmp(fmt, dim == 3 ? "PiiiiiP" : "PiiiP")) return fail(SLF_ERR_INVALID, "arguments: (dist, face, base_gx[, base_other], max_lx[, max_other], buffer)");
      break;
    case KK_COLLECT_MACRO_FACE:
    case KK_DISTRIBUTE_MACRO_FACE:                                // 3-D (field, face, base_gx, base_other, max_lx, max_other, buffer)
      want_p = 2; want_i = dim == 3 ? 5 : 3;                      // 2-D (field, base_gx, max_lx, gy, buffer)   kernel_utils.mako:839-953
      if (strcmp(fmt, dim == 3 ? "PiiiiiP" : "PiiiP")) return fail(SLF_ERR_INVALID, "arguments: (field, [face,] base_gx, ..., buffer)");
      break;
    case KK_SC_MACRO:
    case KK_SC_SWEEP0:
    case KK_SC_SWEEP1: want_p = 5 + dim; want_i = 1; break;       // map, dist, dist, rho, phi, v.., options
    case KK_SC_FUSED: want_p = 7 + dim; want_i = 1; break;        // map, dist0 in, out, dist1 in, out, rho, phi, v.., options
    case KK_SC_INIT: want_p = 5 + dim; want_i = 0; break;         // map, dist1, dist2, v.., rho, phi
    case KK_SCS_MACRO: want_p = 3; want_i = 1; break;             // map, dist, rho, options
    case KK_SCS_SWEEP: want_p = 4 + dim; want_i = 1; break;       // as CollideAndPropagate
    case KK_RESIDENT: want_p = 5; want_i = 5; break;              // map, src a, src b, dst a, dst b, options, steps, tile x, tile y, halo
  }
  if (k->kind == KK_RESIDENT && k->ptrs.size() == 5 && k->ints.size() == 5) {
    const bool aa = k->mod->access_pattern == SLF_AA;
    const int steps = (int)k->ints[1], tx = (int)k->ints[2], ty = (int)k->ints[3], halo = (int)k->ints[4];
    if (!needs_iteration) return fail(SLF_ERR_INVALID, "CollideAndPropagateResident needs the iteration argument (its first step)");
    if (steps < 1 || tx < 1 || ty < 1 || halo < 1) return fail(SLF_ERR_INVALID, "steps, tile and halo must be positive");
    if (!k->ptrs[1] || !k->ptrs[3] || (!aa && (!k->ptrs[2] || !k->ptrs[4])))
      return fail(SLF_ERR_INVALID, "CollideAndPropagateResident: source and destination arrays (both copies of the two-copy pattern)");
    if (k->ptrs[1] == k->ptrs[3] || (!aa && k->ptrs[2] == k->ptrs[4]))
      return fail(SLF_ERR_INVALID, "CollideAndPropagateResident: source and destination must be different buffers");
    // the halo that `steps` steps need depends on the parity of the first one: the worse of the two must fit
    const int need = std::max(slf::resident_halo(aa, 0, steps), slf::resident_halo(aa, 1, steps));
    if (halo < need) return fail(SLF_ERR_INVALID, "CollideAndPropagateResident: halo too small for this many steps");
    const long long nw = (long long)(tx + 2 * halo) * (long long)(ty + 2 * halo);
    if (nw > 2 * 1024) return fail(SLF_ERR_INVALID, "CollideAndPropagateResident: window larger than 2048 nodes");
    const int q = 9;
    // the window (dynamic LDS) + the kernel's own tables (static: the list of a window's boundary-condition nodes,
    // slf_resident.hip) share the 160 KiB of a workgroup
    if (slf::resident_lds_bytes(q, k->mod->sel.precision, aa, tx + 2 * halo, ty + 2 * halo) + 4352 > 160 * 1024)
      return fail(SLF_ERR_INVALID, "CollideAndPropagateResident: window does not fit the 160 KiB of LDS");
  }
  if (k->mod->geo.indirect && (k->kind == KK_SC_FUSED || k->sc_local_velocity))
    return fail(SLF_ERR_UNSUPPORTED, "indirect addressing: the binary model runs the reference's pass and its two per-lattice sweeps");
  if (k->mod->geo.indirect && (k->kind == KK_COLLIDE_AND_PROPAGATE || k->kind == KK_COMPUTE_MACRO ||
                               k->kind == KK_SET_INITIAL_CONDITIONS || k->kind == KK_SC_MACRO ||
                               k->kind == KK_SC_SWEEP0 || k->kind == KK_SC_SWEEP1 || k->kind == KK_SC_INIT ||
                               k->kind == KK_SCS_MACRO || k->kind == KK_SCS_SWEEP))
    want_p += 1;   // leading `nodes` table (reference _add_indirect_args, subdomain_runner.py:1153-1157)
  if (k->ptrs.size() != want_p || k->ints.size() != want_i)
    return fail(SLF_ERR_INVALID, "argument list does not match the kernel's signature");
  k->needs_iteration = needs_iteration;
  k->bound = true;
  if (k->kind == KK_COLLIDE_AND_PROPAGATE && k->mod->geo.indirect && !k->ptrs.empty() &&
      k->mod->phys.incompressible != SLF_DENSITY_ROUNDOFF && !k->mod->phys.regularized && !k->mod->phys.subgrid) {
    // indirect addressing: the sweep is launched over the slots (slot_sweep_kernel); the slot -> node table of this
    // `nodes` argument is built here, once (a launch may be recorded into a graph, where nothing can be allocated)
    SLF_HIP(hipSetDevice(k->mod->ctx->device));
    slot_table_for(k->mod, (const void*)k->ptrs[0], (hipStream_t)0);
  }
  return SLF_OK;
}

int slf_kernel_set_iteration(slf_kernel* k, uint32_t iteration) {
  if (!k) return fail(SLF_ERR_INVALID, "kernel is NULL");
  k->iteration = iteration;
  return SLF_OK;
}

int slf_kernel_launch(slf_kernel* k, const slf_region* region, slf_stream* stream) {
  if (!k) return fail(SLF_ERR_INVALID, "kernel is NULL");
  if (!k->bound) return fail(SLF_ERR_INVALID, "kernel arguments not set");
  slf_module* m = k->mod;
  const slf::Geometry& g = m->geo;
  SLF_HIP(hipSetDevice(m->ctx->device));   // several contexts may live in one process (one per GPU): launch on ours
  hipStream_t s = native(stream);
  hipError_t e = hipSuccess;
  switch (k->kind) {
    case KK_COLLIDE_AND_PROPAGATE:
    case KK_COMPUTE_MACRO: {
      slf::SweepArgs a = {};
      const int b0 = g.indirect ? 1 : 0;
      a.nodes = g.indirect ? (const void*)k->ptrs[0] : nullptr;
      a.map = (const void*)k->ptrs[b0 + 0];
      a.dist_in = (void*)k->ptrs[b0 + 1];
      a.dist_out = (void*)k->ptrs[b0 + 2];
      a.rho = (void*)k->ptrs[b0 + 3];
      a.phi = nullptr;
      a.v[0] = (void*)k->ptrs[b0 + 4];
      a.v[1] = (void*)k->ptrs[b0 + 5];
      a.v[2] = g.dim == 3 ? (void*)k->ptrs[b0 + 6] : nullptr;
      a.node_params = m->node_params;
      a.status = m->status;
      a.options = (uint32_t)k->ints[0];
      for (int f = 0; f < 2; f++) {
        a.xsend[f] = m->xsend[f];
        a.xrecv[f] = m->xrecv[f];
      }
      a.rows = m->rows.map ? &m->rows : nullptr;
      if (m->sel.general && !a.map) return fail(SLF_ERR_INVALID, "node map is NULL but the module is not fluid_only");
      if (g.indirect && !a.nodes) return fail(SLF_ERR_INVALID, "indirect addressing: the nodes table is NULL");
      // (the slot table is built when the arguments are bound, slf_kernel_set_args: a launch may be part of a graph capture)
      if (g.indirect && k->kind == KK_COLLIDE_AND_PROPAGATE && m->slots.nodes == a.nodes) a.slots = &m->slots;
      slf::Prop prop = slf::PROP_AB;
      if (m->access_pattern == SLF_AA) {
        if (!k->needs_iteration) return fail(SLF_ERR_INVALID, "AA kernels need the iteration argument");
        prop = (k->iteration & 1u) ? slf::PROP_AA_ODD : slf::PROP_AA_EVEN;
      }
      if (k->kind == KK_COMPUTE_MACRO) {
        e = slf::launch_macro(m->sel, prop, g, m->phys, a, s);
        break;
      }
      int y0 = 1, y1 = g.lat_ny - 1, z0 = 1, z1 = g.lat_nz - 1;
      if (g.dim == 2) { z0 = 0; z1 = 1; }
      if (region) {
        y0 = region->y0; y1 = region->y1;
        if (g.dim == 3) { z0 = region->z0; z1 = region->z1; }
        if (y0 < 1 || y1 > g.lat_ny - 1 || y0 > y1 || (g.dim == 3 && (z0 < 1 || z1 > g.lat_nz - 1 || z0 > z1)))
          return fail(SLF_ERR_INVALID, "region outside the real nodes of the subdomain");
      }
      e = slf::launch_sweep(m->sel, prop, g, m->phys, a, y0, y1, z0, z1, m->block_x, s);
      break;
    }
    case KK_SC_MACRO:
    case KK_SC_SWEEP0:
    case KK_SC_SWEEP1: {
      slf::SweepArgs a = {};
      const int b0 = g.indirect ? 1 : 0;     // indirect: (nodes, map, dist, dist, rho, phi, v.., options), lb_binary.py:457-465
      a.nodes = g.indirect ? (const void*)k->ptrs[0] : nullptr;
      if (g.indirect && !a.nodes) return fail(SLF_ERR_INVALID, "indirect addressing: the nodes table is NULL");
      a.map = (const void*)k->ptrs[b0 + 0];
      a.dist_in = (void*)k->ptrs[b0 + 1];
      a.dist_out = (void*)k->ptrs[b0 + 2];
      a.rho = (void*)k->ptrs[b0 + 3];
      a.phi = (void*)k->ptrs[b0 + 4];
      a.v[0] = (void*)k->ptrs[b0 + 5];
      a.v[1] = (void*)k->ptrs[b0 + 6];
      a.v[2] = g.dim == 3 ? (void*)k->ptrs[b0 + 7] : nullptr;
      a.node_params = m->node_params;
      a.status = m->status;
      a.options = (uint32_t)k->ints[0];
      a.sc_local_velocity = k->sc_local_velocity;
      sc_planes_of(m, a);
      if (m->sel.general && !a.map) return fail(SLF_ERR_INVALID, "node map is NULL but the module is not fluid_only");
      slf::Prop prop = slf::PROP_AB;
      if (m->access_pattern == SLF_AA) {
        if (!k->needs_iteration) return fail(SLF_ERR_INVALID, "AA kernels need the iteration argument");
        prop = (k->iteration & 1u) ? slf::PROP_AA_ODD : slf::PROP_AA_EVEN;
      }
      int y0 = 1, y1 = g.lat_ny - 1, z0 = 1, z1 = g.lat_nz - 1;
      if (g.dim == 2) { z0 = 0; z1 = 1; }
      if (region) {
        y0 = region->y0; y1 = region->y1;
        if (g.dim == 3) { z0 = region->z0; z1 = region->z1; }
        if (y0 < 1 || y1 > g.lat_ny - 1 || y0 > y1 || (g.dim == 3 && (z0 < 1 || z1 > g.lat_nz - 1 || z0 > z1)))
          return fail(SLF_ERR_INVALID, "region outside the real nodes of the subdomain");
      }
      if (k->kind == KK_SC_MACRO) e = slf::launch_sc_macro(m->sel, prop, g, m->phys, m->sc, a, y0, y1, z0, z1, s);
      else e = slf::launch_sc_sweep(m->sel, k->kind == KK_SC_SWEEP0 ? 0 : 1, prop, g, m->phys, m->sc, a, y0, y1, z0, z1,
                                    m->block_x, s);
      break;
    }
    case KK_SC_FUSED: {
      slf::SweepArgs a = {};
      a.map = (const void*)k->ptrs[0];
      a.dist_in = (void*)k->ptrs[1];
      a.dist_out = (void*)k->ptrs[2];
      a.dist_in2 = (void*)k->ptrs[3];
      a.dist_out2 = (void*)k->ptrs[4];
      a.rho = (void*)k->ptrs[5];
      a.phi = (void*)k->ptrs[6];
      a.v[0] = (void*)k->ptrs[7];
      a.v[1] = (void*)k->ptrs[8];
      a.v[2] = g.dim == 3 ? (void*)k->ptrs[9] : nullptr;
      a.node_params = m->node_params;
      a.status = m->status;
      a.options = (uint32_t)k->ints[0];
      a.sc_local_velocity = k->sc_local_velocity;
      sc_planes_of(m, a);
      if (m->sel.general && !a.map) return fail(SLF_ERR_INVALID, "node map is NULL but the module is not fluid_only");
      slf::Prop prop = slf::PROP_AB;
      if (m->access_pattern == SLF_AA) {
        if (!k->needs_iteration) return fail(SLF_ERR_INVALID, "AA kernels need the iteration argument");
        prop = (k->iteration & 1u) ? slf::PROP_AA_ODD : slf::PROP_AA_EVEN;
      }
      int y0 = 1, y1 = g.lat_ny - 1, z0 = 1, z1 = g.lat_nz - 1;
      if (g.dim == 2) { z0 = 0; z1 = 1; }
      if (region) {
        y0 = region->y0; y1 = region->y1;
        if (g.dim == 3) { z0 = region->z0; z1 = region->z1; }
        if (y0 < 1 || y1 > g.lat_ny - 1 || y0 > y1 || (g.dim == 3 && (z0 < 1 || z1 > g.lat_nz - 1 || z0 > z1)))
          return fail(SLF_ERR_INVALID, "region outside the real nodes of the subdomain");
      }
      e = slf::launch_sc_fused(m->sel, prop, g, m->phys, m->sc, a, y0, y1, z0, z1, m->block_x, s);
      break;
    }
    case KK_RESIDENT: {
      slf::SweepArgs a = {};
      a.map = (const void*)k->ptrs[0];
      a.node_params = m->node_params;
      a.status = m->status;
      a.options = (uint32_t)k->ints[0] & ~1u;       // no field output from inside a stretch of resident steps
      if (m->sel.general && !a.map) return fail(SLF_ERR_INVALID, "node map is NULL but the module is not fluid_only");
      const void* src[2] = {(const void*)k->ptrs[1], (const void*)k->ptrs[2]};
      void* dst[2] = {(void*)k->ptrs[3], (void*)k->ptrs[4]};
      e = slf::launch_resident(m->sel, m->access_pattern == SLF_AA, g, m->phys, a, src, dst, (int)k->iteration, (int)k->ints[1],
                               (int)k->ints[2], (int)k->ints[3], (int)k->ints[4], s);
      break;
    }
    case KK_SCS_MACRO:
    case KK_SCS_SWEEP: {
      slf::SweepArgs a = {};
      const int b0 = g.indirect ? 1 : 0;     // indirect: the `nodes` table leads the list (reference _add_indirect_args)
      a.nodes = g.indirect ? (const void*)k->ptrs[0] : nullptr;
      if (g.indirect && !a.nodes) return fail(SLF_ERR_INVALID, "indirect addressing: the nodes table is NULL");
      a.map = (const void*)k->ptrs[b0 + 0];
      a.dist_in = (void*)k->ptrs[b0 + 1];
      a.phi = nullptr;
      a.node_params = m->node_params;
      a.status = m->status;
      a.options = (uint32_t)k->ints[0];
      if (k->kind == KK_SCS_MACRO) {
        a.dist_out = nullptr;
        a.rho = (void*)k->ptrs[b0 + 2];
        a.v[0] = a.v[1] = a.v[2] = nullptr;
      } else {
        a.dist_out = (void*)k->ptrs[b0 + 2];
        a.rho = (void*)k->ptrs[b0 + 3];
        a.v[0] = (void*)k->ptrs[b0 + 4];
        a.v[1] = (void*)k->ptrs[b0 + 5];
        a.v[2] = g.dim == 3 ? (void*)k->ptrs[b0 + 6] : nullptr;
      }
      sc_planes_of(m, a);
      if (m->sel.general && !a.map) return fail(SLF_ERR_INVALID, "node map is NULL but the module is not fluid_only");
      slf::Prop prop = slf::PROP_AB;
      if (m->access_pattern == SLF_AA) {
        if (!k->needs_iteration) return fail(SLF_ERR_INVALID, "AA kernels need the iteration argument");
        prop = (k->iteration & 1u) ? slf::PROP_AA_ODD : slf::PROP_AA_EVEN;
      }
      if (k->kind == KK_SCS_MACRO) {
        e = slf::launch_scs_macro(m->sel, prop, g, m->phys, m->sc, a, s);
        break;
      }
      int y0 = 1, y1 = g.lat_ny - 1, z0 = 1, z1 = g.lat_nz - 1;
      if (g.dim == 2) { z0 = 0; z1 = 1; }
      if (region) {
        y0 = region->y0; y1 = region->y1;
        if (g.dim == 3) { z0 = region->z0; z1 = region->z1; }
        if (y0 < 1 || y1 > g.lat_ny - 1 || y0 > y1 || (g.dim == 3 && (z0 < 1 || z1 > g.lat_nz - 1 || z0 > z1)))
          return fail(SLF_ERR_INVALID, "region outside the real nodes of the subdomain");
      }
      e = slf::launch_scs_sweep(m->sel, prop, g, m->phys, m->sc, a, y0, y1, z0, z1, m->block_x, s);
      break;
    }
    case KK_SC_INIT: {
      // ([nodes,] map, dist1, dist2, vx, vy[, vz], rho, phi)  -- reference lb_binary.py:107-127
      const int b0 = g.indirect ? 1 : 0;
      if (g.indirect && !k->ptrs[0]) return fail(SLF_ERR_INVALID, "indirect addressing: the nodes table is NULL");
      const void* v[3] = {(const void*)k->ptrs[b0 + 3], (const void*)k->ptrs[b0 + 4],
                          g.dim == 3 ? (const void*)k->ptrs[b0 + 5] : nullptr};
      e = slf::launch_sc_init(m->sel, g, m->phys, (void*)k->ptrs[b0 + 1], (void*)k->ptrs[b0 + 2],
                              (const void*)k->ptrs[b0 + 3 + g.dim], (const void*)k->ptrs[b0 + 4 + g.dim], v,
                              g.indirect ? (const void*)k->ptrs[0] : nullptr, s);
      break;
    }
    case KK_SET_INITIAL_CONDITIONS: {
      // (dist, vx, vy[, vz], rho, map)  -- reference lb_single.py:72-94
      const int b0 = g.indirect ? 1 : 0;   // indirect: (nodes, dist, v.., rho, map)
      const void* v[3] = {(const void*)k->ptrs[b0 + 1], (const void*)k->ptrs[b0 + 2],
                          g.dim == 3 ? (const void*)k->ptrs[b0 + 3] : nullptr};
      if (g.indirect && !k->ptrs[0]) return fail(SLF_ERR_INVALID, "indirect addressing: the nodes table is NULL");
      e = slf::launch_init(m->sel, g, m->phys, (void*)k->ptrs[b0 + 0], (const void*)k->ptrs[b0 + 1 + g.dim], v,
                           g.indirect ? (const void*)k->ptrs[0] : nullptr, s);
      break;
    }
    case KK_PBC:
    case KK_PBC_SWAP:
      e = slf::launch_pbc(m->sel, g, (void*)k->ptrs[0], (int)k->ints[0], k->kind == KK_PBC_SWAP, s);
      break;
    case KK_MACRO_PBC:
      e = slf::launch_macro_pbc(m->sel, g, (void*)k->ptrs[0], (int)k->ints[0], s);
      break;
    case KK_COLLECT_BOX:
    case KK_DISTRIBUTE_BOX:
      if (k->ints.size() >= 6) {
        int dirs[32], nd = 0;
        for (int q = 0; q < 32; q++) if (((unsigned int)k->ints[0] >> q) & 1u) dirs[nd++] = q;    // ascending
        if (nd > 12) return fail(SLF_ERR_INVALID, "Collect/DistributeContinuousData: at most 12 directions per launch (a face of D3Q19 carries 5)");
        e = slf::launch_box(m->sel, g, k->kind == KK_COLLECT_BOX, (void*)k->ptrs[0], (void*)k->ptrs[1], dirs, nd,
                            (unsigned long long)(uint32_t)k->ints[1], (long long)k->ints[2], (int)k->ints[3],
                            (long long)k->ints[4], (int)k->ints[5], k->ints.size() == 8 ? (long long)k->ints[6] : 0,
                            k->ints.size() == 8 ? (long long)k->ints[7] : 0, false, s);
        break;
      }
      [[fallthrough]];
    case KK_COLLECT_FACE_SWAP:
    case KK_DISTRIBUTE_FACE_SWAP:
    case KK_COLLECT_MACRO_FACE:
    case KK_DISTRIBUTE_MACRO_FACE: {
      // The reference's face kernels (kernel_utils.mako:476-953), mapped onto the box kernel.  face: 2 Y_LOW, 3 Y_HIGH,
      // 4 Z_LOW, 5 Z_HIGH (subdomain.py:30-36); the layer a kernel reads / writes is lat_linear / lat_linear_macro /
      // lat_linear_dist / lat_linear_with_swap of subdomain_runner.py:486-510; the populations are
      // get_interblock_dists(grid, normal(face)) in ascending order -- their opposite slots in the ...WithSwap kernels --
      // buffer [k][other][x] (2-D: [k][x]).
      const bool collect = k->kind == KK_COLLECT_BOX || k->kind == KK_COLLECT_FACE_SWAP || k->kind == KK_COLLECT_MACRO_FACE;
      const bool swap = k->kind == KK_COLLECT_FACE_SWAP || k->kind == KK_DISTRIBUTE_FACE_SWAP;
      const bool macro = k->kind == KK_COLLECT_MACRO_FACE || k->kind == KK_DISTRIBUTE_MACRO_FACE;
      int face, base_gx, base_other = 0, max_lx, max_other = 1, layer = -1;
      if (g.dim == 3) {
        face = (int)k->ints[0]; base_gx = (int)k->ints[1]; base_other = (int)k->ints[2];
        max_lx = (int)k->ints[3]; max_other = (int)k->ints[4];
      } else if (macro) {
        face = 2; base_gx = (int)k->ints[0]; max_lx = (int)k->ints[1]; layer = (int)k->ints[2];
      } else {
        face = (int)k->ints[0]; base_gx = (int)k->ints[1]; max_lx = (int)k->ints[2];
      }
      if (face < 2 || face >= 2 * g.dim) return fail(SLF_ERR_INVALID, "face must be a Y or Z face (X faces travel through index lists / x-face buffers)");
      const int axis = face >> 1;
      const bool low = (face & 1) == 0;
      const int lat = axis == 1 ? g.lat_ny : g.lat_nz;
      if (layer < 0) {
        if (macro) layer = collect ? (low ? 1 : lat - 2) : (low ? 0 : lat - 1);             // lat_linear_macro / lat_linear
        else if (collect) layer = swap ? (low ? 1 : lat - 2) : (low ? 0 : lat - 1);        // lat_linear_macro / lat_linear
        else layer = swap ? (low ? lat - 1 : 0) : (low ? lat - 2 : 1);                     // lat_linear_with_swap / lat_linear_dist
      }
      int dirs[32], nd = 0;
      if (macro) {
        dirs[nd++] = 0;
      } else {
        const int sign = low ? -1 : 1;
        for (int q = 1; q < (g.dim == 3 ? 19 : 9); q++) {
          const int ea = g.dim == 3 ? slf::e_comp<slf::D3Q19>(q, axis) : slf::e_comp<slf::D2Q9>(q, axis);
          if (ea == sign) dirs[nd++] = swap ? (g.dim == 3 ? slf::D3Q19::opp(q) : slf::D2Q9::opp(q)) : q;
        }
      }
      int ncols, nrows;
      if (g.dim == 3) {
        if (max_other % nd) return fail(SLF_ERR_INVALID, "max_other must be a multiple of the number of populations that cross the face");
        ncols = max_lx; nrows = max_other / nd;
      } else {
        if (!macro && max_lx % nd) return fail(SLF_ERR_INVALID, "max_lx must be a multiple of the number of populations that cross the face");
        ncols = macro ? max_lx : max_lx / nd; nrows = 1;
      }
      const long long row_stride = axis == 1 ? (long long)g.arr_nxy : (long long)g.arr_nx;   // rows run along the other axis
      const unsigned long long base = (unsigned long long)base_gx +
          (axis == 1 ? (unsigned long long)g.arr_nx * layer + (unsigned long long)g.arr_nxy * base_other
                     : (unsigned long long)g.arr_nx * base_other + (unsigned long long)g.arr_nxy * layer);
      const int lat_other = g.dim == 3 ? (axis == 1 ? g.lat_nz : g.lat_ny) : 1;       // extent along the rows' axis
      if (base_gx < 0 || base_gx + ncols > g.arr_nx || base_other < 0 || base_other + nrows > lat_other || layer < 0 || layer >= lat)
        return fail(SLF_ERR_INVALID, "face box outside the subdomain (base_gx + max_lx, base_other + rows or the layer exceed the arrays)");
      slf::Geometry gm = g;
      if (macro) gm.dist_size = 0;
      e = slf::launch_box(m->sel, gm, collect, (void*)k->ptrs[0], (void*)k->ptrs[1], dirs, nd, base, 1, ncols, row_stride, nrows,
                          0, 0, macro, s);
      break;
    }
    case KK_COLLECT_SPARSE:
    case KK_DISTRIBUTE_SPARSE:
      e = slf::launch_sparse(m->sel, k->kind == KK_COLLECT_SPARSE, (const unsigned long long*)k->ptrs[0],
                             (void*)k->ptrs[1], (void*)k->ptrs[2], (int)k->ints[0], s);
      break;
  }
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  return SLF_OK;
}

// ---- step plans: the launch list of one time step, built once, enqueued with ONE call ----------------------------
// (include/sailfish_hip.h "step plans".)  Every entry is what the corresponding slf_* call would do; slf_plan_run()
// walks the list on the calling thread -- no Python, no argument marshalling between the entries.
namespace {
enum PlanKind { PL_LAUNCH, PL_RECORD, PL_WAIT, PL_EXCHANGE, PL_MEMSET, PL_COPY, PL_XFACE, PL_PEER_SIGNAL, PL_PEER_WAIT };
struct PlanOp {
  PlanKind kind;
  slf_kernel* k = nullptr;
  bool has_region = false;
  slf_region region = {0, 0, 0, 0};
  slf_stream* stream = nullptr;
  slf_event* ev = nullptr;
  slf_comm* comm = nullptr;
  std::vector<slf_comm_op> ops;
  void* dst = nullptr;
  const void* src = nullptr;
  int value = 0;
  size_t bytes = 0;
  slf_module* mod = nullptr;
  void* xf[4] = {nullptr, nullptr, nullptr, nullptr};
  slf_peer* peer = nullptr;
  std::vector<int32_t> ranks;
  int channel = 0;
};
}  // namespace

struct slf_plan {
  slf_ctx* ctx;
  std::vector<PlanOp> ops;
};

int slf_plan_create(slf_ctx* ctx, slf_plan** out) {
  if (!ctx || !out) return fail(SLF_ERR_INVALID, "NULL argument");
  slf_plan* p = new slf_plan;
  p->ctx = ctx;
  *out = p;
  return SLF_OK;
}

int slf_plan_destroy(slf_plan* p) {
  delete p;
  return SLF_OK;
}

int slf_plan_size(slf_plan* p, int* n_ops) {
  if (!p || !n_ops) return fail(SLF_ERR_INVALID, "NULL argument");
  *n_ops = (int)p->ops.size();
  return SLF_OK;
}

int slf_plan_add_launch(slf_plan* p, slf_kernel* k, const slf_region* region, slf_stream* stream) {
  if (!p || !k) return fail(SLF_ERR_INVALID, "NULL argument");
  if (!k->bound) return fail(SLF_ERR_INVALID, "kernel arguments not set");
  PlanOp o;
  o.kind = PL_LAUNCH;
  o.k = k;
  o.has_region = region != nullptr;
  if (region) o.region = *region;
  o.stream = stream;
  p->ops.push_back(o);
  return SLF_OK;
}

int slf_plan_add_record(slf_plan* p, slf_event* ev, slf_stream* stream) {
  if (!p || !ev) return fail(SLF_ERR_INVALID, "NULL argument");
  PlanOp o;
  o.kind = PL_RECORD;
  o.ev = ev;
  o.stream = stream;
  p->ops.push_back(o);
  return SLF_OK;
}

int slf_plan_add_wait(slf_plan* p, slf_stream* stream, slf_event* ev) {
  if (!p || !ev) return fail(SLF_ERR_INVALID, "NULL argument");
  PlanOp o;
  o.kind = PL_WAIT;
  o.ev = ev;
  o.stream = stream;
  p->ops.push_back(o);
  return SLF_OK;
}

int slf_plan_add_exchange(slf_plan* p, slf_comm* comm, const slf_comm_op* ops, int n, slf_stream* stream) {
  if (!p || !comm || (!ops && n > 0)) return fail(SLF_ERR_INVALID, "NULL argument");
  for (int i = 0; i < n; i++) {
    if (ops[i].elem_bytes != 4 && ops[i].elem_bytes != 8 && ops[i].elem_bytes != 1)
      return fail(SLF_ERR_INVALID, "elem_bytes must be 1, 4 or 8");
    if (ops[i].kind != SLF_COMM_SEND && ops[i].kind != SLF_COMM_RECV) return fail(SLF_ERR_INVALID, "bad slf_comm_op kind");
  }
  PlanOp o;
  o.kind = PL_EXCHANGE;
  o.comm = comm;
  o.ops.assign(ops, ops + (n > 0 ? n : 0));
  o.stream = stream;
  p->ops.push_back(o);
  return SLF_OK;
}

int slf_plan_add_memset(slf_plan* p, void* dptr, int value, size_t bytes, slf_stream* stream) {
  if (!p || !dptr) return fail(SLF_ERR_INVALID, "NULL argument");
  if (!stream) return fail(SLF_ERR_INVALID, "plan entries are asynchronous: a stream is needed");
  PlanOp o;
  o.kind = PL_MEMSET;
  o.dst = dptr;
  o.value = value;
  o.bytes = bytes;
  o.stream = stream;
  p->ops.push_back(o);
  return SLF_OK;
}

int slf_plan_add_copy(slf_plan* p, void* dst, const void* src, size_t bytes, slf_stream* stream) {
  if (!p || !dst || !src) return fail(SLF_ERR_INVALID, "NULL argument");
  if (!stream) return fail(SLF_ERR_INVALID, "plan entries are asynchronous: a stream is needed");
  PlanOp o;
  o.kind = PL_COPY;
  o.dst = dst;
  o.src = src;
  o.bytes = bytes;
  o.stream = stream;
  p->ops.push_back(o);
  return SLF_OK;
}

int slf_plan_add_xface_buffers(slf_plan* p, slf_module* m, void* send_low, void* send_high, void* recv_low, void* recv_high) {
  if (!p || !m) return fail(SLF_ERR_INVALID, "NULL argument");
  // validated now, with the module's rules, so that slf_plan_run cannot fail on it
  if (int e = check_xface_buffers(m, send_low, send_high, recv_low, recv_high)) return e;
  PlanOp o;
  o.kind = PL_XFACE;
  o.mod = m;
  o.xf[0] = send_low; o.xf[1] = send_high; o.xf[2] = recv_low; o.xf[3] = recv_high;
  p->ops.push_back(o);
  return SLF_OK;
}

int slf_plan_add_xface_planes(slf_plan* p, slf_module* m, int32_t which, void* send_low, void* send_high, void* recv_low,
                              void* recv_high) {
  if (!p || !m) return fail(SLF_ERR_INVALID, "NULL argument");
  if (int e = check_xface_planes(m, which, send_low, send_high, recv_low, recv_high)) return e;
  PlanOp o;
  o.kind = PL_XFACE;
  o.mod = m;
  o.value = which + 1;        // 0: the single-fluid buffers
  o.xf[0] = send_low; o.xf[1] = send_high; o.xf[2] = recv_low; o.xf[3] = recv_high;
  p->ops.push_back(o);
  return SLF_OK;
}

static int plan_add_peer(slf_plan* p, PlanKind kind, slf_peer* peer, const int32_t* ranks, int n, int channel, int count,
                         slf_stream* stream) {
  if (!p) return fail(SLF_ERR_INVALID, "NULL argument");
  if (count < 1) return fail(SLF_ERR_INVALID, "peer wait: count must be at least 1");
  // validated now, with the transport's rules, so that slf_plan_run cannot fail on it
  if (int e = peer_check_ranks(peer, ranks, n, channel)) return e;
  if (!stream) return fail(SLF_ERR_INVALID, "plan entries are asynchronous: a stream is needed");
  PlanOp o;
  o.kind = kind;
  o.peer = peer;
  o.ranks.assign(ranks, ranks + n);
  o.channel = channel;
  o.value = count;
  o.stream = stream;
  p->ops.push_back(o);
  return SLF_OK;
}

int slf_plan_add_peer_signal(slf_plan* p, slf_peer* peer, const int32_t* ranks, int n, int channel, slf_stream* stream) {
  return plan_add_peer(p, PL_PEER_SIGNAL, peer, ranks, n, channel, 1, stream);
}

int slf_plan_add_peer_wait(slf_plan* p, slf_peer* peer, const int32_t* ranks, int n, int channel, int count, slf_stream* stream) {
  return plan_add_peer(p, PL_PEER_WAIT, peer, ranks, n, channel, count, stream);
}

int slf_plan_run(slf_plan* p, uint32_t iteration) {
  if (!p) return fail(SLF_ERR_INVALID, "plan is NULL");
  for (PlanOp& o : p->ops) {
    int e = SLF_OK;
    switch (o.kind) {
      case PL_LAUNCH:
        if (o.k->needs_iteration) o.k->iteration = iteration;
        e = slf_kernel_launch(o.k, o.has_region ? &o.region : nullptr, o.stream);
        break;
      case PL_RECORD: e = slf_event_record(o.ev, o.stream); break;
      case PL_WAIT: e = slf_stream_wait_event(o.stream, o.ev); break;
      case PL_EXCHANGE: e = slf_comm_exchange(o.comm, o.ops.data(), (int)o.ops.size(), o.stream); break;
      case PL_MEMSET: e = slf_memset(p->ctx, o.dst, o.value, o.bytes, o.stream); break;
      case PL_COPY: e = slf_memcpy_d2d_async(p->ctx, o.dst, o.src, o.bytes, o.stream); break;
      case PL_XFACE:
        if (o.value > 0) {
          o.mod->sc_send[o.value - 1][0] = o.xf[0]; o.mod->sc_send[o.value - 1][1] = o.xf[1];
          o.mod->sc_recv[o.value - 1][0] = o.xf[2]; o.mod->sc_recv[o.value - 1][1] = o.xf[3];
          break;
        }
        o.mod->xsend[0] = o.xf[0]; o.mod->xsend[1] = o.xf[1];
        o.mod->xrecv[0] = o.xf[2]; o.mod->xrecv[1] = o.xf[3];
        break;
      case PL_PEER_SIGNAL: e = slf_peer_signal(o.peer, o.ranks.data(), (int)o.ranks.size(), o.channel, o.stream); break;
      case PL_PEER_WAIT: e = slf_peer_wait(o.peer, o.ranks.data(), (int)o.ranks.size(), o.channel, o.value, o.stream); break;
    }
    if (e) return e;
  }
  return SLF_OK;
}

}  // extern "C"
