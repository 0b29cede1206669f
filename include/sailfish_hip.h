/* sailfish_hip.h -- C ABI of libsailfish_hip.so, the MI355X (gfx950) backend
 * for the Sailfish Lattice-Boltzmann hot path.
 *
 * This is the drop-in boundary: the reference's plugin seam is the *backend
 * module* (sailfish/util.py:52-59 get_backends; instantiated per subdomain
 * process at sailfish/master.py:46-49).  Each entry point below states which
 * method of the reference's backend interface (sailfish/backend_cuda.py,
 * identical in backend_opencl.py) it replaces.  sailfish_amd/backend_hip.py
 * binds these with ctypes and exposes exactly the reference's Python-side
 * contract; INTEGRATION.md shows the binding.
 *
 * Conventions: plain C types only; every function returns 0 on success and a
 * non-zero status otherwise (the message is available from slf_last_error());
 * device pointers are passed as void* (64-bit device addresses); all launches
 * and *_async copies are asynchronous on the given stream (NULL = the
 * context's default stream); no callbacks; no global mutable state other than
 * the per-thread last-error string.
 */
#ifndef SAILFISH_HIP_H
#define SAILFISH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLF_ABI_VERSION 1

typedef struct slf_ctx slf_ctx;
typedef struct slf_stream slf_stream;
typedef struct slf_event slf_event;
typedef struct slf_module slf_module;
typedef struct slf_kernel slf_kernel;

enum { SLF_OK = 0, SLF_ERR_INVALID = 1, SLF_ERR_HIP = 2, SLF_ERR_UNSUPPORTED = 3, SLF_ERR_NOT_FOUND = 4 };

enum { SLF_D2Q9 = 0, SLF_D3Q19 = 1 };
enum { SLF_BGK = 0, SLF_MRT = 1 };
enum { SLF_AB = 0, SLF_AA = 1 };
/* density model = the value of slf_module_desc::incompressible (reference sym.py:573-661, sym_equilibrium.py:100-118):
 * compressible (rho0 = rho), --incompressible (rho0 = 1), --minimize_roundoff (the arrays hold f_i - w_i, the density
 * variable is rho - 1, rho0 = rho; BGK with fluid and bounce-back nodes only, as in the reference "BGK-like models") */
enum { SLF_DENSITY_COMPRESSIBLE = 0, SLF_DENSITY_INCOMPRESSIBLE = 1, SLF_DENSITY_ROUNDOFF = 2 };
enum { SLF_SIM_LBM = 0, SLF_SIM_SHAN_CHEN_BINARY = 1, SLF_SIM_SHAN_CHEN_SINGLE = 2 };

/* Node addressing (reference subdomain_runner.py:829-878, kernel_common.mako:140-167).  INDIRECT: the
 * distribution arrays hold the *active* nodes only (dist_stride >= number of active nodes); a dense
 * uint32 table `nodes` maps a node's dense index to its slot or SLF_INVALID_NODE.  Node map and
 * macroscopic fields stay dense.  In an indirect module CollideAndPropagate, SetInitialConditions and
 * ComputeMacroFields -- and, for SLF_SIM_SHAN_CHEN_BINARY, ShanChenPrepareMacroFields and
 * ShanChenCollideAndPropagate0 / 1 (lb_binary.py:121-123, 457-465; the fused sweep is not offered), for
 * SLF_SIM_SHAN_CHEN_SINGLE PrepareMacroFields as well -- take that table as an additional FIRST pointer argument (as
 * the reference's _add_indirect_args, subdomain_runner.py:1153-1157); periodic boundaries must be wrapped in-sweep. */
enum { SLF_ADDR_DIRECT = 0, SLF_ADDR_INDIRECT = 1 };

/* How a body-force / Shan-Chen acceleration a enters the BGK collision (relaxation_common.mako:56-99):
 * GUO: feq at u + a/2 plus Guo's source term; EDM (exact difference method): feq at u, then
 * f_i += feq_i(rho, u + a) - feq_i(rho, u).  The output velocity is u + a/2 in both. */
enum { SLF_FORCE_GUO = 0, SLF_FORCE_EDM = 1 };
enum { SLF_SUBGRID_NONE = 0, SLF_SUBGRID_LES_SMAGORINSKY = 1 };
#define SLF_INVALID_NODE 0xffffffffu

/* Canonical node kinds understood by the kernels (reference node_type.py:86-109,
 * 115-168, 198-, 269-; wet/excluded semantics from templates/geo_helpers.mako:42-84). */
enum {
  SLF_NK_FLUID = 0,
  SLF_NK_GHOST = 1,
  SLF_NK_UNUSED = 2,
  SLF_NK_PROPAGATION_ONLY = 3,
  SLF_NK_FULL_BB = 4,
  SLF_NK_HALF_BB = 5,
  SLF_NK_REGULARIZED_VELOCITY = 6,
  SLF_NK_EQUILIBRIUM_DENSITY = 7,
  SLF_NK_EQUILIBRIUM_VELOCITY = 8,
  SLF_NK_ZOUHE_VELOCITY = 9,      /* boundary.mako:343-382, 811-815 */
  SLF_NK_ZOUHE_DENSITY = 10,      /* boundary.mako:487-494 */
  SLF_NK_REGULARIZED_DENSITY = 11, /* boundary.mako:501-506, 817-835 */
  SLF_NK_COPY = 12,               /* NTCopy, boundary.mako:574-587: unknown populations copied from the node
                                     one step along the inward normal (two-copy access pattern only) */
  SLF_NK_YU_OUTFLOW = 13,         /* NTYuOutflow, boundary.mako:589-603: 2 f(x + n) - f(x + 2 n) (AB only) */
  SLF_NK_DO_NOTHING = 14,         /* NTDoNothing, boundary.mako:862-876: in-place (AA) pattern -- the unknown populations keep
                                     their value: the node stores them where its next step reads them (its own slot in an odd
                                     step, the opposite slot of the ghost node behind it in an even one); two-copy pattern: a
                                     plain fluid node (node_type.py:296-307); refused with indirect addressing (ghost nodes own
                                     no slot) */
  SLF_NK_SLIP = 15                /* NTSlip, boundary.mako:837-855 + sym.py:481-497: dry node, specular reflection -- every
                                     population with a component along the normal swaps with its mirror image */
};

#define SLF_MAX_NODE_TYPES 16
#define SLF_MAX_Q 27

/* Replaces the *source text* handed to backend.build() (reference
 * sailfish/codegen.py:104-180 renders Mako -> CUDA C; backend_cuda.py:193-218
 * compiles it).  Kernels here are pre-built for gfx950, so build() receives
 * this descriptor: the same information the reference puts in its template
 * context (subdomain_runner.py:181-247 update_context, lb_base.py:120-137,
 * geo_encoder.py:341-363). */
typedef struct slf_module_desc {
  uint32_t struct_size;        /* sizeof(slf_module_desc), ABI check */
  int32_t lattice;             /* SLF_D2Q9 | SLF_D3Q19            (--grid) */
  int32_t model;               /* SLF_BGK | SLF_MRT               (--model) */
  int32_t precision;           /* 4 (single) | 8 (double)          (--precision) */
  int32_t access_pattern;      /* SLF_AB | SLF_AA                  (--access_pattern) */
  int32_t lat_nx, lat_ny, lat_nz; /* logical size incl. the ghost envelope; lat_nz = 1 in 2-D */
  int32_t arr_nx, arr_ny, arr_nz; /* padded in-memory size (subdomain_runner.py:359-373) */
  int32_t envelope;            /* ghost layer width, always 1 (controller.py:482-494) */
  int32_t periodic_fused[3];   /* 1: the sweep wraps this axis itself (no ghost traffic, no PBC kernel) */
  int32_t incompressible;      /* density model SLF_DENSITY_*: 0, 1 = --incompressible, 2 = --minimize_roundoff */
  int32_t relaxation_enabled;  /* 0 = streaming only (regtest propagation KATs) */
  int32_t has_force;           /* body force present (lb_base.py:331-359) */
  int32_t fluid_only;          /* 1: no node map is read; every real node is a fluid node */
  double tau;                  /* (6 visc + 1)/2, sym.py:847-848 */
  double visc;
  double accel[3];             /* body-force acceleration */
  double mrt_rates[SLF_MAX_Q]; /* per-moment relaxation rates (sym.py:78-149, 331-406) */
  /* node-code bit fields, geo_encoder.py:365-382: orientation | scratch | param | type */
  uint32_t nt_type_mask;
  uint32_t nt_misc_shift;      /* bits of the type field */
  uint32_t nt_param_shift;     /* bits of the param-index field */
  uint32_t nt_scratch_shift;   /* bits of the scratch-id field */
  int32_t n_types;
  int32_t type_kind[SLF_MAX_NODE_TYPES]; /* dense type id -> SLF_NK_* */
  int32_t use_link_tags;       /* orientation field of half-BB nodes holds link tags (subdomain.py:593-642) */
  int32_t n_node_params;
  const double* node_params;   /* kernel_common.mako:523-536; copied at module creation */
  uint64_t dist_stride;        /* elements between consecutive direction arrays; 0 = arr_nx*arr_ny*arr_nz (the
                                  reference's DIST_SIZE, kernel_common.mako:506).  A larger stride only inserts
                                  unused padding between the Q arrays (HBM channel de-aliasing). */
  int32_t periodic_local[3];   /* 1: the subdomain spans this (globally periodic) axis, i.e. it is its own periodic
                                  neighbour there (SubdomainSpec.enable_local_periodicity, subdomain.py:122-127);
                                  0: the faces of this axis are walls / open / connected to other subdomains */
  int32_t simtype;             /* SLF_SIM_LBM | SLF_SIM_SHAN_CHEN_BINARY (lb_binary.py:375-517) |
                                  SLF_SIM_SHAN_CHEN_SINGLE (lb_single.py:242-347; coupling constant in sc_G[0]) */
  /* binary Shan-Chen: relaxation time of the second lattice (--tau_phi), coupling constants
   * G11, G12, G21, G22 (lb_binary.py:388-394) and pseudopotential (sym.py:896-908: 0 linear, 1 classic) */
  double tau_phi;
  double sc_G[4];
  int32_t sc_potential;
  int32_t node_addressing;       /* SLF_ADDR_DIRECT | SLF_ADDR_INDIRECT (--node_addressing, lb_base.py:66-71) */
  double accel1[3];              /* binary models: body-force acceleration acting on lattice 1 (accel[] acts on
                                    lattice 0; reference add_body_force(..., grid=1), lb_base.py:331-359) */
  int32_t force_implementation;  /* SLF_FORCE_GUO | SLF_FORCE_EDM (--force_implementation, lb_base.py:325-330) */
  int32_t sparse_geometry;       /* hint: a sizeable fraction of the real nodes is excluded (unused / ghost-like):
                                    kernels predicate their loads on the node map instead of issuing them early */
  /* round 6: the two options of the BGK relaxation preamble (relaxation_common.mako:166-237) no example of the
   * reference uses.  regularized (--regularized, lb_single.py:27-30): before the collision the populations are replaced
   * by feq + the projection of their non-equilibrium momentum flux (sym.reglb_flux_tensor).  subgrid = SLF_SUBGRID_*
   * (--subgrid=les-smagorinsky, lb_single.py:38-42): the relaxation time of a node follows the local non-equilibrium
   * stress, tau0 = 1/2 + 3 visc, tau0 += (sqrt(tau0^2 + 36 C^2 sqrt(T_ab T_ab)) - tau0) / 2 with C = smagorinsky_const.
   * Single-fluid BGK modules, standard density models, Guo or no body force; served by the per-node kernels. */
  int32_t regularized;
  int32_t subgrid;
  double smagorinsky_const;
} slf_module_desc;

/* Region of the lattice a sweep launch covers (replaces the reference's
 * bulk/boundary grid arithmetic, subdomain_runner.py:396-475 +
 * kernel_common.mako:242-457).  Rows y in [y0, y1), planes z in [z0, z1);
 * x always spans the whole row.  NULL region = every real node. */
typedef struct slf_region {
  int32_t y0, y1, z0, z1;
} slf_region;

/* ---- context / memory: backend.__init__, alloc_buf, alloc_async_host_buf,
 *      to_buf/from_buf(_async)  (backend_cuda.py:67-191) ---- */
int slf_abi_version(void);
int slf_device_count(int* count);
int slf_device_pci_bus_id(int device, char* out, size_t len);    /* "0000:05:00.0": which NUMA node the GPU hangs off
                                                                    (the controller pins each subdomain process to
                                                                    cores of its GPU's node; reference master.py:106-117
                                                                    only deals the GPUs out) */
int slf_ctx_create(int device, slf_ctx** out);
int slf_ctx_destroy(slf_ctx* ctx);
int slf_ctx_sync(slf_ctx* ctx);                                  /* backend.sync() */
int slf_ctx_info(slf_ctx* ctx, char* name, size_t name_len, size_t* total_mem, int* cu_count,
                 int* wavefront);                                /* backend.info / total_memory / get_defines */
int slf_ctx_free_memory(slf_ctx* ctx, size_t* free_bytes);       /* device memory not in use by ANY process / context
                                                                    (sizes the span of placed arrays) */
int slf_malloc(slf_ctx* ctx, size_t bytes, void** dptr);         /* alloc_buf */
int slf_free(slf_ctx* ctx, void* dptr);
int slf_memset(slf_ctx* ctx, void* dptr, int value, size_t bytes, slf_stream* stream);
int slf_host_alloc_pinned(size_t bytes, void** hptr);            /* alloc_async_host_buf */
int slf_host_free(void* hptr);
int slf_memcpy_h2d(slf_ctx* ctx, void* dptr, const void* hptr, size_t bytes);        /* to_buf */
int slf_memcpy_d2h(slf_ctx* ctx, void* hptr, const void* dptr, size_t bytes);        /* from_buf */
int slf_memcpy_h2d_async(slf_ctx* ctx, void* dptr, const void* hptr, size_t bytes, slf_stream* s); /* to_buf_async */
int slf_memcpy_d2h_async(slf_ctx* ctx, void* hptr, const void* dptr, size_t bytes, slf_stream* s); /* from_buf_async */
int slf_memcpy_d2d_async(slf_ctx* ctx, void* dst, const void* src, size_t bytes, slf_stream* s);
int slf_memcpy_peer_async(slf_ctx* ctx, void* dst, int dst_device, const void* src, int src_device,
                          size_t bytes, slf_stream* s);          /* same-process multi-GPU halo */

/* ---- placed allocations (extension; no counterpart in the reference).  One virtual address range, backed by
 *      separately created physical chunks that can be swapped under a fixed address.  Why: the sweep keeps 2 x Q
 *      streams open in HBM at once, and on MI355X its rate depends on WHICH physical regions those streams fall
 *      into (5.3 TB/s when the whole distribution array lies inside one ~16 GiB region, 6.4 TB/s when it straddles
 *      two: profiles/r02/README.md); a plain hipMalloc lands in either case at random.  The backend therefore
 *      reserves the range and backs it with 16 chunks created alternately with spacer allocations, which spreads
 *      them over ~70 GiB of physical HBM; the runner then times a few steps of the real kernels, places once more
 *      while the first chunks stay allocated and keeps the better placement (sailfish_amd/placement.py: place(),
 *      tune()).  All sizes / addresses are multiples of slf_vmm_granularity. ---- */
int slf_vmm_granularity(slf_ctx* ctx, size_t* bytes);
int slf_vmm_reserve(slf_ctx* ctx, size_t bytes, void** va);              /* address range only, no memory */
int slf_vmm_release_range(slf_ctx* ctx, void* va, size_t bytes);
int slf_vmm_chunk_create(slf_ctx* ctx, size_t bytes, uint64_t* handle); /* physical memory, not yet addressable */
int slf_vmm_chunk_release(slf_ctx* ctx, uint64_t handle);
int slf_vmm_map(slf_ctx* ctx, void* va, size_t bytes, uint64_t handle);  /* whole chunk at va, read-write */
int slf_vmm_unmap(slf_ctx* ctx, void* va, size_t bytes);

/* ---- halo exchange between subdomain processes: replaces SubdomainRunner._send_dists / _recv_dists + the zmq
 *      connectors (reference subdomain_runner.py:1064-1139, connector.py:73-174: device -> pinned host -> socket
 *      -> pinned host -> device) by RCCL point-to-point operations between device buffers (xGMI inside a node).
 *      One communicator per context; rank = subdomain process.  The 128-byte id of slf_comm_unique_id() is created
 *      by one rank and handed to the others by the host code (any side channel).  slf_comm_sendrecv() posts one
 *      send and / or one receive with `peer` on `stream` (counts in elements of elem_bytes = 1 | 4 | 8); exchanges
 *      with several peers in one step go between slf_comm_group_begin() and slf_comm_group_end().  RCCL is bound at
 *      run time (dlopen), so a process that never creates a communicator does not need it.  sailfish_amd's own
 *      runner reaches the same RCCL through torch.distributed (backend "nccl"); these entry points are the
 *      boundary for a host that does not carry torch. ---- */
typedef struct slf_comm slf_comm;
int slf_comm_unique_id(void* id128);
int slf_comm_init(slf_ctx* ctx, int nranks, int rank, const void* unique_id, slf_comm** out);
int slf_comm_destroy(slf_comm* comm);
/* Size of the communicator and this process's rank in it as RCCL reports them (ncclCommCount, ncclCommUserRank);
 * either pointer may be NULL.  bench.py's `rccl_ranks` is this number (0 when the halos do not travel over RCCL). */
int slf_comm_count(slf_comm* comm, int* nranks, int* rank);
int slf_comm_group_begin(void);
int slf_comm_group_end(void);
int slf_comm_sendrecv(slf_comm* comm, int peer, const void* send_dptr, size_t n_send, void* recv_dptr, size_t n_recv,
                      int elem_bytes, slf_stream* stream);
/* The whole batch of a halo exchange in one call: n sends / receives posted in the given order (the order in which a
 * pair of ranks matches them) inside one RCCL group on `stream`. */
enum { SLF_COMM_SEND = 0, SLF_COMM_RECV = 1 };
typedef struct slf_comm_op {
  int32_t kind;        /* SLF_COMM_SEND | SLF_COMM_RECV */
  int32_t peer;
  void* dptr;
  uint64_t count;      /* elements */
  int32_t elem_bytes;  /* 1 | 4 | 8 */
  int32_t reserved;
} slf_comm_op;
int slf_comm_exchange(slf_comm* comm, const slf_comm_op* ops, int n, slf_stream* stream);

/* ---- peer transport (round 6): halo buffers of one subdomain process mapped into its neighbours' address space.
 *      The reference left the hook for it -- backend.ipc_handle / ipc_handle_wrap (backend_cuda.py:122-126), unused --
 *      and moves its halos device -> host -> socket -> host -> device (connector.py:73-174).  On one node every GPU
 *      reaches every other one's memory over xGMI, so the kernels that PRODUCE a halo (edge lanes of the sweep, the
 *      pack kernels) store straight into the receive buffers of the neighbouring process, and what is left of the
 *      exchange is ordering: a progress counter per (sender, receiver, channel) that the sender advances with a
 *      one-lane kernel behind its writes and the receiver's stream waits for with a one-lane kernel -- no copy, no
 *      RCCL kernels, nothing on the host.  Works between processes that SHARE a device as well (HIP IPC maps the same
 *      physical memory), which is what makes the multi-process step measurable on a single-GPU box.
 *        slf_peer_create      the progress counters of this process (uncached device memory), one block per process;
 *        slf_peer_flags_handle / slf_peer_connect   hand the block's IPC handle to the other ranks (any side channel;
 *                             64 opaque bytes) / map theirs; the own rank needs no connect;
 *        slf_peer_alloc       a device buffer other processes can map + its 64-byte handle; slf_peer_open maps a buffer
 *                             of another process (whole allocation; offsets are the caller's business);
 *        slf_peer_signal      after everything enqueued on `stream` so far (its writes released to system scope):
 *                             counter[me -> r][channel] = ++sent[r][channel] in the memory of every listed rank r;
 *        slf_peer_wait        `stream` continues when `count` further signals of every listed rank r have arrived:
 *                             counter[r -> me][channel] >= (awaited[r][channel] += count).  Waits and signals of a
 *                             pair match by order on the channel: both sides enqueue the same sequence per pair, as
 *                             they must for RCCL; one wait may cover several signals (count > 1).
 *      A wait gives up after the time-out (slf_peer_set_timeout, default 60 s; the spin is bounded so that a lost
 *      neighbour cannot hang the device) and leaves {1, rank, channel, expected, seen} in slf_peer_status, which the
 *      host polls when it synchronises anyway.  Write-after-read is the caller's: a receive buffer must not be written
 *      again before its reader is done -- sailfish_amd alternates two buffer sets by step parity and orders the rest
 *      through the same counters (xface.ChunkPlan.peer_need).
 *      slf_peer_selftest_* : fill a (mapped) buffer with a pattern / count the words of a local buffer that differ from
 *      it -- the transport checks at start-up, on the real mappings, that plain device memory written through a mapping
 *      and released by a signal is what the waiting process reads (dense and one-word-per-line stores). ---- */
typedef struct slf_peer slf_peer;
enum { SLF_PEER_HANDLE_BYTES = 64, SLF_PEER_CHANNELS = 4 };
int slf_peer_create(slf_ctx* ctx, int nranks, int rank, slf_peer** out);
int slf_peer_destroy(slf_peer* peer);
int slf_peer_flags_handle(slf_peer* peer, void* handle64);
int slf_peer_connect(slf_peer* peer, int rank, const void* handle64);
int slf_peer_alloc(slf_peer* peer, size_t bytes, void** dptr, void* handle64);
int slf_peer_free(slf_peer* peer, void* dptr);
int slf_peer_open(slf_peer* peer, const void* handle64, void** mapped);
int slf_peer_close(slf_peer* peer, void* mapped);
int slf_peer_signal(slf_peer* peer, const int32_t* ranks, int n, int channel, slf_stream* stream);
int slf_peer_wait(slf_peer* peer, const int32_t* ranks, int n, int channel, int count, slf_stream* stream);
int slf_peer_set_timeout(slf_peer* peer, double seconds);
/* out = {timed out (0 | 1), rank waited for, channel, expected count, count seen, waits timed out so far, 0, 0} */
int slf_peer_status(slf_peer* peer, int64_t out[8]);
/* what this process has enqueued / seen for one (rank, channel): signals sent to it, signals awaited from it, and the
 * counter it has advanced here so far (read from the device: synchronises the NULL stream only) */
int slf_peer_progress(slf_peer* peer, int rank, int channel, uint64_t* sent, uint64_t* awaited, uint64_t* arrived);
int slf_peer_selftest_fill(slf_peer* peer, void* dst, size_t nwords, uint32_t pattern, int one_word_per_block,
                           slf_stream* stream);
int slf_peer_selftest_check(slf_peer* peer, const void* src, size_t nwords, uint32_t pattern, slf_stream* stream,
                            uint32_t* bad_words);                      /* waits for `stream` */

/* ---- streams / events: make_stream, make_event, sync_stream
 *      (backend_cuda.py:291-308, 24-52) ---- */
int slf_stream_create(slf_ctx* ctx, slf_stream** out);
int slf_stream_create_high_priority(slf_ctx* ctx, slf_stream** out);  /* the halo stream (subdomain_runner.py:825-827 has two
                                                                       equal streams; CUDA / HIP offer a priority) */
int slf_stream_destroy(slf_stream* s);
int slf_stream_sync(slf_stream* s);
int slf_stream_native(slf_stream* s, void** hip_stream);         /* raw hipStream_t, for torch.cuda.ExternalStream */
int slf_stream_wait_event(slf_stream* s, slf_event* ev);         /* stream.wait_for_event */
int slf_event_create(slf_ctx* ctx, int timing, slf_event** out);
int slf_event_destroy(slf_event* ev);
int slf_event_record(slf_event* ev, slf_stream* s);
int slf_event_sync(slf_event* ev);
int slf_event_elapsed_ms(slf_event* start, slf_event* end, float* ms); /* ev.time_since */

/* ---- launch graphs (extension; no counterpart in the reference, which pays one Python-level launch per
 *      kernel: subdomain_runner.py:960-974).  Everything enqueued on `s` between begin and end -- kernel
 *      launches, copies, event waits -- is recorded instead of executed; the result replays with ONE
 *      runtime call.  The runner uses it for launch-bound small subdomains (2-D cases): stretches of
 *      steps without host interaction are replayed as graphs of 2..16 steps. ---- */
typedef struct slf_graph slf_graph;
int slf_graph_capture_begin(slf_stream* s);
int slf_graph_capture_end(slf_stream* s, slf_graph** out);
int slf_graph_launch(slf_graph* g, slf_stream* s);
int slf_graph_destroy(slf_graph* g);

/* ---- modules / kernels: build, get_kernel, set_iteration, run_kernel
 *      (backend_cuda.py:193-251, 128-130) ---- */
int slf_module_create(slf_ctx* ctx, const slf_module_desc* desc, slf_module** out); /* build() */
int slf_module_destroy(slf_module* m);
/* On-GPU invalid value check (reference --check_invalid_results_gpu, geo_helpers.mako:193-213): sweep kernels
 * launched with bit 2 (value 4) of their `options` argument set flag wet nodes whose density is not finite.
 * Waits for `stream`, returns out = {flag, x, y, z} of the first such node and clears the flag. */
int slf_module_poll_invalid(slf_module* m, slf_stream* stream, int32_t out[4]);
/* Boundary-condition parameters that change with time (round 6).  The reference renders time- and space-dependent node
 * parameters into device code (node_type.py:471-626 DynamicValue, LinearlyInterpolatedTimeSeries; boundary.mako:52-84
 * timeseries_interpolate, get_time_from_iteration); the pre-built kernels read constants from the module's parameter
 * table (slf_module_desc::node_params), so the host evaluates such values and rewrites their table entries before
 * every step: entries first .. first + n - 1 take values[] (converted to the module's precision) ON `stream` -- the
 * launches enqueued on that stream afterwards see the new values, those enqueued before it the old ones; the call
 * returns at once (up to 256 values travel as arguments of a one-wave kernel, longer updates through pinned staging
 * buffers the module owns).  first + n must not exceed n_node_params. */
int slf_module_update_node_params(slf_module* m, int first, const double* values, int n, slf_stream* stream);
/* A body force that changes with time (add_body_force(DynamicValue(...)) of sym.S.time, reference lb_base.py:335-353 +
 * the rendered expression in relaxation_common.mako:body_force): the acceleration is an argument of every sweep launch,
 * so the host sets the value the NEXT launches take -- accel[3] for lattice 0, or lattice 1 of a binary model.  The
 * module must have been created with a body force (has_force, resp. accel1): which instantiation runs was decided
 * there.  Forces that depend on position are not served (they would need a field). */
int slf_module_set_body_force(slf_module* m, int lattice, const double accel[3]);
/* Row classes of the node map at `map_dptr` (no counterpart in the reference, whose kernels decode the map in every
 * thread: geo_helpers.mako:146-161, kernel_common.mako:191-201).  Builds, on the device, one class per 64-node
 * x-segment of every row -- 0 = all plain fluid: the wavefront that owns the segment neither reads the map (4 of the
 * 156 bytes a D3Q19 f32 update moves) nor runs node-type code -- and the list of rows that hold boundary-condition
 * nodes; CollideAndPropagate launches whose map argument is `map_dptr` then sweep the listed rows with the
 * instantiation for the module's node-type table and all other rows with the small fluid / bounce-back one.
 * out_counts (may be NULL) = {rows, rows with boundary-condition nodes, segments, plain-fluid segments}.
 * Optional; call again when the map's contents change, with map_dptr = NULL to drop the tables.  D3Q19 modules with
 * a node map and direct addressing.  Waits for `stream`. */
int slf_module_classify_rows(slf_module* m, const void* map_dptr, slf_stream* stream, int32_t out_counts[4]);
/* Kernel names are the reference's: "CollideAndPropagate",
 * "SetInitialConditions", "ApplyPeriodicBoundaryConditions",
 * "ApplyPeriodicBoundaryConditionsWithSwap" (AA modules only),
 * "ApplyMacroPeriodicBoundaryConditions", "CollectSparseData", "DistributeSparseData"
 * (index lists are uint64: q * dist_stride + node index),
 * "CollectContinuousData" / "DistributeContinuousData"(dist, buffer, dirs, base, col_stride, ncols, row_stride, nrows
 * [, buffer stride between directions, buffer stride between rows])
 * (reference kernel_utils.mako:526-543, 629-645, 692-708, 777-793: face planes without index lists -- the
 * populations in bit mask `dirs` of the node box base + c col_stride + r row_stride <-> dense buffer [k][r][c];
 * 'i' arguments, base < 2^32; at most 12 directions per launch -- the mask travels to the kernel as an ordered list
 * of 5-bit entries; a face of D3Q19 carries 5, more is SLF_ERR_INVALID), plus "ComputeMacroFields"
 * (rho / v of the current state, arguments as CollideAndPropagate).
 * The reference's own face kernels are served under their names AND argument lists as well (a host that binds the
 * reference's _init_collect_kernels / _init_distrib_kernels, subdomain_runner.py:1160-1290, needs no special case):
 * "CollectContinuousData" / "DistributeContinuousData" / "CollectContinuousDataWithSwap" /
 * "DistributeContinuousDataWithSwap"(dist, face, base_gx, base_other, max_lx, max_other, buffer), format "PiiiiiP"
 * [2-D: (dist, face, base_gx, max_lx, buffer), "PiiiP"] and "CollectContinuousMacroData" /
 * "DistributeContinuousMacroData"(field, face, base_gx, base_other, max_lx, max_other, buffer) [2-D: (field, base_gx,
 * max_lx, gy, buffer)] -- kernel_utils.mako:476-953: face = 2 Y_LOW, 3 Y_HIGH, 4 Z_LOW, 5 Z_HIGH; the populations are
 * get_interblock_dists(grid, normal(face)) in ascending order (their OPPOSITE slots in the ...WithSwap kernels, which
 * exist in AA modules only); the layer read / written is the reference's lat_linear (Collect: ghost layer of `face`),
 * lat_linear_macro (CollectWithSwap, CollectMacro: first real layer), lat_linear_dist (Distribute: first real layer of
 * the far side), lat_linear_with_swap (DistributeWithSwap: ghost layer of the far side), lat_linear (DistributeMacro),
 * subdomain_runner.py:486-510; buffer [k][other][x], max_other = rows x populations; a box that leaves the arrays
 * (base_gx + max_lx beyond the padded row, base_other + rows beyond the other axis) is SLF_ERR_INVALID.  Deviation: the reference strides
 * the rows of the macro buffers by its launch grid's x size (kernel_utils.mako:886, 940); launch geometry is not part
 * of this boundary, the rows are max_lx apart (dense).  The two forms of Collect / DistributeContinuousData are told
 * apart by their argument formats.  x faces travel through index lists (or the x-face buffers below) as in the
 * reference (subdomain_runner.py:1294-1306).
 * Binary Shan-Chen modules (simtype = SLF_SIM_SHAN_CHEN_BINARY) provide instead of CollideAndPropagate:
 * "ShanChenPrepareMacroFields"(map, dist1, dist2, rho, phi, vx, vy[, vz], options),
 * "ShanChenCollideAndPropagate0|1"(map, dist_in, dist_out, rho, phi, vx, vy[, vz], options),
 * "ShanChenCollideAndPropagateFused"(map, dist0_in, dist0_out, dist1_in, dist1_out, rho, phi, vx, vy[, vz], options)
 * = both of them in one pass (the fields and the pseudopotential stencil are read once; same results),
 * the pair "ShanChenPrepareDensities" (arguments of ShanChenPrepareMacroFields) + "ShanChenCollideAndPropagateFusedV"
 * (arguments of ...Fused) = the same step with the node's own densities and the common velocity formed inside the
 * sweep from the populations it loads anyway (same operation order, same bits): the pass in front stores rho and phi
 * only, and vx, vy, vz only in launches whose options have bit 0 set (the reference's full_output kernels,
 * lb_binary.py:434-440) -- between two such launches the velocity arrays keep the values of the last one; and the
 * two-lattice "SetInitialConditions"(map, dist1, dist2, vx, vy[, vz], rho, phi)
 * (reference templates/models/binary_shan_chen.mako:19-141, lb_binary_fluid.mako:87-127).
 * Single-component Shan-Chen modules (SLF_SIM_SHAN_CHEN_SINGLE) keep the single-fluid kernel names; their
 * "CollideAndPropagate" adds the pseudopotential force, and "PrepareMacroFields"(map, dist, rho, options)
 * computes the density field it reads (reference templates/models/lb_single_fluid.mako:129-229).
 * Launch-bound 2-D subdomains (no counterpart in the reference, which launches CollideAndPropagate once per step,
 * subdomain_runner.py:960-974): "CollideAndPropagateResident"(map, src_a, src_b, dst_a, dst_b, options, steps, tile_x,
 * tile_y, halo), format "PPPPPiiiii", needs_iteration = 1 (the iteration of its FIRST step) performs `steps` time
 * steps in one launch: every workgroup keeps a window (tile + halo) of the RAW distribution arrays in LDS, steps it
 * exactly as CollideAndPropagate steps memory (same node code, same slots; even / odd in-place iterations resp. the two
 * copies alternating) and writes its tile back -- every slot of every node, ghost layer included.  src_a / dst_a: the
 * array of the in-place pattern resp. copy A of the two-copy pattern, src_b / dst_b: copy B (NULL in place); source and
 * destination must be different buffers that agree outside the lattice box (padding, ghost columns of an axis wrapped
 * in-sweep).  halo >= steps + 1 (two-copy) resp. 2 ceil(steps / 2) (in place); (tile + 2 halo)^2 <= 2048 nodes and
 * 160 KiB of LDS.  SLF_ERR_UNSUPPORTED for 3-D, Shan-Chen, indirect-addressing and --minimize_roundoff modules, axes
 * periodic through the ghost-layer kernels, and type tables with half-way bounce-back or outflow nodes (their node code
 * reads / writes memory itself); options bit 0 (field output) is ignored. */
int slf_kernel_get(slf_module* m, const char* name, slf_kernel** out);
int slf_kernel_destroy(slf_kernel* k);
/* fmt: one char per argument, 'P' = device pointer (8 bytes), 'i' = int32,
 * 'f' = float, 'd' = double (reference lb_single.py:122 struct-style formats).
 * argv[i] points at the value.  Argument order = the reference kernel's.
 * needs_iteration: a trailing uint32 iteration counter is appended and
 * rewritten by slf_kernel_set_iteration (backend_cuda.py:241-245). */
int slf_kernel_set_args(slf_kernel* k, const char* fmt, const void* const* argv, int argc,
                        int needs_iteration);
int slf_kernel_set_iteration(slf_kernel* k, uint32_t iteration);
int slf_kernel_launch(slf_kernel* k, const slf_region* region, slf_stream* stream); /* run_kernel */

/* x faces connected to another subdomain (1-D decompositions along x, the reference's default axis, geo.py:100-135):
 * instead of pushing into the ghost columns x = 0 / nx + 1 and packing them with a strided gather afterwards
 * (reference Collect/DistributeContinuousData on an x face, kernel_utils.mako:526-543), the two edge lanes of every
 * row write what leaves the subdomain straight into send_* and read what enters it from recv_* -- dense buffers
 * [z][k][y] of arr_nz * 5 * arr_ny reals per face (k = rank of the direction among those with e_x > 0 for the high
 * face / the values entering through the low face, e_x < 0 otherwise; rows cover the padded plane; a range of
 * z-planes is one contiguous piece, so the planes a z-chunk of the sweep has completed can travel while the next
 * chunk computes).  Each step the host only moves send_high -> the high neighbour's recv_low and send_low -> the low
 * neighbour's recv_high.  Entries whose bits are all ones (memset 0xFF) mean "nothing crossed the face here" and are
 * ignored by the reader -- fill ALL FOUR buffers that way before the first step: the arrays then count; every
 * other value, NaN and infinities included, is delivered.  A send buffer needs no clearing between steps as long as
 * each buffer only ever serves one kind of step (in place: one set for the even, one for the odd iterations; two-copy:
 * any): which entries a step writes is fixed by the node map, the others keep their all-ones for good.  The send buffer of
 * one subdomain may BE the receive buffer of its neighbour (same device: nothing to move).  The pointers may be changed between launches (the
 * runner alternates two sets by step parity).  All CollideAndPropagate kernels of the module use the buffers once set; NULL pointers
 * switch a face back to ghost columns.  D3Q19 single-fluid modules, direct addressing, x not wrapped in-sweep. */
int slf_module_set_xface_buffers(slf_module* m, void* send_low, void* send_high, void* recv_low, void* recv_high);

/* The same for the binary Shan-Chen model (two lattices coupled through the densities their force stencil reads at the
 * neighbouring nodes; reference lb_binary.py:393-517 with subdomain_runner.py:1907-2197 _send_macro / _recv_macro and
 * the ghost-column blocks of both lattices).  Three sets of planes per connected face, chosen by `which`:
 *   0, 1  the populations of lattice 0 / 1: layout, markers and ownership rules of slf_module_set_xface_buffers;
 *   2     the densities: [z][field][y] over the padded plane, arr_nz * 2 * arr_ny reals per face (field 0 = rho,
 *         1 = phi).  "ShanChenPrepareDensities" stores rho and phi of the subdomain's first / last column into
 *         send_low / send_high; the edge lanes of "ShanChenCollideAndPropagateFusedV" take the five stencil values per
 *         field that lie across the face from recv_low / recv_high (the neighbour's send_high / send_low of the SAME
 *         step) and never touch the ghost columns.  No markers: an entry is either rewritten every step (a node whose
 *         density the pass forms) or never (a node the pass skips, a ghost row) -- FILL every entry of the send planes
 *         once from the fields, before the first step and after every host-side write of the state
 *         (`CollectContinuousData` on column x = 1 / nx of rho and phi: xface.NNPlanes.prime); a wet edge node in a row
 *         next to a y / z face that is neither wrapped in-sweep nor a wall reads ghost-row entries, whose content is the
 *         caller's business as it is with ghost columns.
 * The single-component model (`PrepareMacroFields` / `CollideAndPropagate`, reference lb_single.py:242-347) takes sets 0
 * and 2 (field 0 of the density planes; same size).
 * A connected face needs all of its sets.  For D3Q19 modules with direct addressing, x not wrapped inside the sweep; with
 * or without a node map; both access patterns (in place: the even step stores what the neighbour's odd step pulls into
 * the own row of the plane, as the single-fluid kernels do); only the kernels named above run with planes set (the
 * others refuse). */
int slf_module_set_xface_planes(slf_module* m, int32_t which, void* send_low, void* send_high, void* recv_low,
                                void* recv_high);

/* The ghost columns x = 0 (low) / x = nx + 1 (high) of this subdomain carry nothing the simulation uses -- the face is
 * a wall or open, neither periodic through the ghost-layer kernels nor connected to another subdomain through ghost
 * columns, and (in-place pattern) the first / last real column holds no wet node: the odd in-place step of a WET node
 * next to an open face pulls out of the ghost column what its even step pushed there, so such a face must keep its
 * ghost column; what a dry (bounce-back) node pulls out of it only ever travels back into it.  The whole-row sweeps
 * then neither push into these columns (five partial-line writes per row and face) nor pull out of them (five lines
 * per row and face fetched for one value each).  The reference pushes into ghost nodes regardless
 * (propagation.mako:384-421); only raw dumps of the arrays can tell the difference.  The caller decides (it owns the
 * node map): sailfish_amd/subdomain_runner.py _init_compute. */
int slf_module_set_x_ghost_unused(slf_module* m, int low, int high);

/* number of x-threads per workgroup the sweep uses for this module (diagnostics) */
int slf_module_block_size(slf_module* m, int* threads);

/* ---- step plans (extension): the launch list of ONE time step, built once, enqueued with one call.
 *      The reference enqueues every kernel, event and copy of a step from Python (SubdomainRunner.step(),
 *      subdomain_runner.py:960-974; the boundary / bulk overlap with its event chain, :1028-1058; _send_dists /
 *      _recv_dists, :1064-1139) -- fine at its 15 ms per step, most of a 0.9 ms MI355X step once a halo-connected
 *      subdomain needs ~20 runtime calls per step (z-chunks of the sweep, an event and an RCCL batch after each).
 *      A plan holds that list: kernel launches with their regions and streams, event records, stream waits, RCCL
 *      batches (the ops are copied), buffer fills and device-to-device copies, and the x-face buffer set the sweeps use
 *      from there on.  slf_plan_run(plan, iteration) performs the entries in order on the calling thread; kernels bound
 *      with needs_iteration get `iteration` (as slf_kernel_set_iteration) before they are launched.  Kernels, events,
 *      streams, communicators and modules named in a plan must outlive it.  Events recorded by a plan may be waited
 *      for by another plan or by direct calls (a wait refers to the most recent record at the time it is enqueued;
 *      waiting for an event that was never recorded is a no-op).  A runner keeps one plan per (step parity, with /
 *      without field output): four plans cover every step of a run.  On error the run stops at the failing entry. ---- */
typedef struct slf_plan slf_plan;
int slf_plan_create(slf_ctx* ctx, slf_plan** out);
int slf_plan_destroy(slf_plan* plan);
int slf_plan_size(slf_plan* plan, int* n_ops);
int slf_plan_add_launch(slf_plan* plan, slf_kernel* k, const slf_region* region, slf_stream* stream);   /* run_kernel */
int slf_plan_add_record(slf_plan* plan, slf_event* ev, slf_stream* stream);                             /* make_event */
int slf_plan_add_wait(slf_plan* plan, slf_stream* stream, slf_event* ev);                               /* wait_for_event */
int slf_plan_add_exchange(slf_plan* plan, slf_comm* comm, const slf_comm_op* ops, int n, slf_stream* stream);
int slf_plan_add_memset(slf_plan* plan, void* dptr, int value, size_t bytes, slf_stream* stream);
int slf_plan_add_copy(slf_plan* plan, void* dst, const void* src, size_t bytes, slf_stream* stream);    /* same device */
int slf_plan_add_xface_buffers(slf_plan* plan, slf_module* m, void* send_low, void* send_high, void* recv_low,
                               void* recv_high);                                /* slf_module_set_xface_buffers */
int slf_plan_add_xface_planes(slf_plan* plan, slf_module* m, int32_t which, void* send_low, void* send_high,
                              void* recv_low, void* recv_high);                 /* slf_module_set_xface_planes */
int slf_plan_add_peer_signal(slf_plan* plan, slf_peer* peer, const int32_t* ranks, int n, int channel, slf_stream* stream);
int slf_plan_add_peer_wait(slf_plan* plan, slf_peer* peer, const int32_t* ranks, int n, int channel, int count,
                           slf_stream* stream);
int slf_plan_run(slf_plan* plan, uint32_t iteration);

const char* slf_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SAILFISH_HIP_H */
