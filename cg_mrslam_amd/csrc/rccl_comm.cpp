// RCCL all-gather of the condensed-graph wire buffers over xGMI (include/cgmr.h, "exchange").
//
// Reference behaviour being replaced: GraphComm's pairwise UDP send / receive of ComboMessages
// (src/mrslam/graph_comm.cpp:103-208: sendToThrd serialises one message per robot in range, receiveFromThrd
// deserialises whatever arrives) -- here ONE collective per round: every rank contributes its fixed-capacity wire
// buffer, every rank receives all of them and reads the slices addressed to it.
//
// librccl is resolved at run time (dlopen): inside a PyTorch process the copy PyTorch already loaded is reused (two
// RCCL copies in one process must not both initialise), a plain C++ process gets the system library.  The collective
// runs on the communicator's own side stream, ordered after the caller's stream by an event, so that the next
// optimize() on the context's stream overlaps with it; cgmr_comm_wait() orders the context's stream
// after the collective without blocking the host.
#include <dlfcn.h>

#include <cstring>

#include "cgmr_ctx.h"

using namespace cgmr;

namespace {

// the slice of rccl.h this file needs (ncclResult_t 0 = success; ncclUint8 = 1; ncclUniqueId = 128 bytes)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef int (*fn_get_unique_id)(ncclUniqueId_t*);
typedef int (*fn_comm_init_rank)(ncclComm_t*, int, ncclUniqueId_t, int);
typedef int (*fn_comm_destroy)(ncclComm_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
typedef const char* (*fn_get_error_string)(int);
typedef int (*fn_comm_count)(const ncclComm_t, int*);

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_get_error_string get_error_string = nullptr;
  fn_comm_count comm_count = nullptr, comm_user_rank = nullptr;      // (diagnostics only: may be absent)
  bool ok = false;
};

Rccl& rccl() {
  static Rccl R = [] {
    Rccl r;
    // 1. a copy that is already in the process (PyTorch's), 2. the library path the caller names, 3. the system one
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!r.handle && getenv("CGMR_RCCL_LIB")) r.handle = dlopen(getenv("CGMR_RCCL_LIB"), RTLD_NOW | RTLD_GLOBAL);
    for (const char* n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) return r;
    r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.all_gather = (fn_all_gather)dlsym(r.handle, "ncclAllGather");
    r.get_error_string = (fn_get_error_string)dlsym(r.handle, "ncclGetErrorString");
    r.comm_count = (fn_comm_count)dlsym(r.handle, "ncclCommCount");
    r.comm_user_rank = (fn_comm_count)dlsym(r.handle, "ncclCommUserRank");
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather;
    return r;
  }();
  return R;
}

}  // namespace

struct cgmr_comm {
  cgmr_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int n_ranks = 1, rank = 0;
  hipStream_t stream = nullptr;          // side stream of the collective
  hipEvent_t ready = nullptr, done = nullptr;
  double last_seconds = 0;
  hipEvent_t t0 = nullptr, t1 = nullptr;
};

extern "C" {

// librccl resolves in this process (every symbol the exchange needs): what a rank that is not the root of the unique id
// checks before the collective initialisation -- ncclGetUniqueId on every rank would start a bootstrap listener per rank
// that nothing ever uses.
int cgmr_comm_probe(void) { return rccl().ok ? CGMR_OK : CGMR_E_NO_DEVICE; }

int cgmr_comm_unique_id(void* id_out_128) {
  if (!id_out_128) return CGMR_E_INVALID;
  Rccl& R = rccl();
  if (!R.ok) return CGMR_E_NO_DEVICE;
  ncclUniqueId_t id;
  if (R.get_unique_id(&id) != 0) return CGMR_E_HIP;
  memcpy(id_out_128, &id, 128);
  return CGMR_OK;
}

int cgmr_comm_create(cgmr_ctx* ctx, int n_ranks, int rank, const void* unique_id_128, cgmr_comm** out) {
  if (!ctx || !out || !unique_id_128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return CGMR_E_INVALID;
  *out = nullptr;
  Rccl& R = rccl();
  if (!R.ok) return set_err(ctx, CGMR_E_NO_DEVICE, "librccl could not be loaded (set CGMR_RCCL_LIB)");
  if (hipSetDevice(ctx->device) != hipSuccess) return set_err(ctx, CGMR_E_NO_DEVICE, "hipSetDevice failed");
  cgmr_comm* c = new cgmr_comm();
  c->ctx = ctx; c->n_ranks = n_ranks; c->rank = rank;
  ncclUniqueId_t id;
  memcpy(&id, unique_id_128, 128);
  int rc = R.comm_init_rank(&c->comm, n_ranks, id, rank);
  if (rc != 0) {
    set_err(ctx, CGMR_E_HIP, "ncclCommInitRank: %s", R.get_error_string ? R.get_error_string(rc) : "error");
    delete c;
    return CGMR_E_HIP;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess || hipEventCreate(&c->t0) != hipSuccess ||
      hipEventCreate(&c->t1) != hipSuccess) {
    set_err(ctx, CGMR_E_HIP, "stream / event creation failed");
    cgmr_comm_destroy(c);
    return CGMR_E_HIP;
  }
  *out = c;
  return CGMR_OK;
}

void cgmr_comm_destroy(cgmr_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().comm_destroy(c->comm);
  for (hipEvent_t e : {c->ready, c->done, c->t0, c->t1}) if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// recv[r * bytes_per_rank ...] = rank r's send buffer, for every r.  Issued on the communicator's side stream after
// everything queued so far on the context's stream; returns without waiting.
int cgmr_allgather_condensed(cgmr_ctx* ctx, cgmr_comm* comm, const void* d_send, size_t bytes_per_rank, void* d_recv) {
  if (!ctx || !comm || !d_send || !d_recv || comm->ctx != ctx) return CGMR_E_INVALID;
  Rccl& R = rccl();
  if (hipSetDevice(ctx->device) != hipSuccess) return set_err(ctx, CGMR_E_NO_DEVICE, "hipSetDevice failed");
  if (hipEventRecord(comm->ready, ctx->stream) != hipSuccess || hipStreamWaitEvent(comm->stream, comm->ready, 0) != hipSuccess)
    return set_err(ctx, CGMR_E_HIP, "event ordering failed");
  // ... and behind the context's side stream: a batch of condensed graphs that was not waited for, the message packed behind it
  if (side_join_stream(ctx, comm->stream) != 0) return CGMR_E_HIP;
  (void)hipEventRecord(comm->t0, comm->stream);
  int rc = R.all_gather(d_send, d_recv, bytes_per_rank, /*ncclUint8*/ 1, comm->comm, comm->stream);
  if (rc != 0) return set_err(ctx, CGMR_E_HIP, "ncclAllGather: %s", R.get_error_string ? R.get_error_string(rc) : "error");
  (void)hipEventRecord(comm->t1, comm->stream);
  if (hipEventRecord(comm->done, comm->stream) != hipSuccess) return set_err(ctx, CGMR_E_HIP, "hipEventRecord failed");
  return CGMR_OK;
}

// Everything queued on the context's stream after this call runs after the last all-gather (no host wait).
int cgmr_comm_wait(cgmr_ctx* ctx, cgmr_comm* comm) {
  if (!ctx || !comm || comm->ctx != ctx) return CGMR_E_INVALID;
  if (hipStreamWaitEvent(ctx->stream, comm->done, 0) != hipSuccess) return set_err(ctx, CGMR_E_HIP, "hipStreamWaitEvent failed");
  return CGMR_OK;
}

int cgmr_ctx_join_side(cgmr_ctx* ctx) {
  if (!ctx) return CGMR_E_INVALID;
  return side_join_stream(ctx, ctx->stream);
}

// Device time of the last all-gather (HIP events on the side stream); blocks until it has finished.
int cgmr_comm_last_seconds(cgmr_comm* comm, double* seconds) {
  if (!comm || !seconds) return CGMR_E_INVALID;
  if (hipEventSynchronize(comm->t1) != hipSuccess) return CGMR_E_HIP;
  float ms = 0;
  if (hipEventElapsedTime(&ms, comm->t0, comm->t1) != hipSuccess) return CGMR_E_HIP;
  *seconds = 1e-3 * ms;
  return CGMR_OK;
}

// What the communicator itself says it is (ncclCommCount / ncclCommUserRank): out[0] = ranks, out[1] = this rank's index,
// out[2] = 1 if both came from librccl, 0 if the library lacks the queries and the values are those given to cgmr_comm_create.
// For a multi-GPU run's own report: a rank whose communicator does not span the whole world says so.
int cgmr_comm_info(cgmr_comm* comm, int32_t out[3]) {
  if (!comm || !out) return CGMR_E_INVALID;
  out[0] = comm->n_ranks; out[1] = comm->rank; out[2] = 0;
  Rccl& R = rccl();
  int n = -1, r = -1;
  if (R.comm_count && R.comm_user_rank && comm->comm && R.comm_count(comm->comm, &n) == 0 && R.comm_user_rank(comm->comm, &r) == 0) {
    out[0] = n; out[1] = r; out[2] = 1;
  }
  return CGMR_OK;
}

}  // extern "C"
