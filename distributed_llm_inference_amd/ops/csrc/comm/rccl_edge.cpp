// Native stage-to-stage hop over RCCL (xGMI between the GPUs of one node).
//
// One Edge = one directed pipeline hop (stage i -> i+1, or the token feedback from the last stage
// to stage 0) = its own RCCL communicator + its own HIP stream. The reference's hop
// is an HTTP POST of the whole hidden-state sequence as JSON text, relayed by the orchestrator
// (orchestration.py:114-137; Worker1.py:208-245); here it is ncclSend / ncclRecv of the new rows
// only, enqueued without ever blocking the host:
//
//   send:  fence event on the producing stream -> comm stream waits for it -> ncclSend -> done event
//   recv:  (comm stream waits until the buffer's previous reader has finished) -> ncclRecv -> done
//          event; the consuming stream waits for that event
//
// Every edge owns its communicator and stream, so a forward hop can never queue behind the
// feedback hop, and ncclCommAbort of one edge does not disturb the others. The *_on variants issue
// the transfer directly on a caller's stream, which makes the hop capturable into the hipGraph of
// a decode step (recv -> layers -> send as one graph).
//
// RCCL is resolved at run time from the library PyTorch has already loaded (its path is handed to
// dlia_comm_init): two RCCL / HIP runtimes in one process must never meet. Plain C ABI, used
// through ctypes and by the native stage loop.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <thread>

#include "rccl_edge.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;              // 0 = success, 7 = in progress
constexpr int kNcclInt8 = 0;           // ncclDataType_t: bytes

struct Api {
  ncclResult_t (*GetVersion)(int*);
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommAbort)(ncclComm_t);
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*);
  const char* (*GetErrorString)(ncclResult_t);
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  void* lib = nullptr;
} g_api;

std::mutex g_mu;
thread_local char t_err[512];

int fail(const char* what, const char* detail, int code) {
  snprintf(t_err, sizeof(t_err), "%s: %s (%d)", what, detail ? detail : "?", code);
  return -1;
}
#define HIPCHK(expr)                                                        \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess) return fail(#expr, hipGetErrorString(e_), (int)e_); \
  } while (0)
#define NCCLCHK(expr)                                                       \
  do {                                                                      \
    ncclResult_t r_ = (expr);                                               \
    if (r_ != 0) return fail(#expr, g_api.GetErrorString ? g_api.GetErrorString(r_) : "rccl", (int)r_); \
  } while (0)

constexpr int kSlots = 256;            // completion events in flight per edge (a ring; power of two)
constexpr uint32_t kTokenMask = 0x3FFFFFFFu;   // tokens are transfer sequence numbers modulo 2^30
constexpr int kRingWaitMs = 30000;     // longest a new transfer waits for the one 256 transfers earlier

struct Edge {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, peer = 0, device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t done[kSlots];
  hipEvent_t fence[kSlots];
  bool used[kSlots] = {};
  uint32_t next = 0;                   // sequence number of the next transfer (modulo 2^30)
  std::atomic<int> aborted{0};
  uint64_t bytes_sent = 0, bytes_recv = 0, ops = 0, ring_waits = 0;
};

struct DeviceScope {                   // HIP's current device is per thread; leave it as found
  int prev = -1; bool changed = false;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceScope() { if (changed) hipSetDevice(prev); }
};

// A completion token is the transfer's sequence number; its events live in slot token % kSlots. The
// slot of transfer t is reused by transfer t + kSlots, and re-recording an event somebody still
// waits for would make that waiter observe the NEWER transfer: a new transfer therefore takes its
// slot only once the transfer that last used it has completed (the host polls with a deadline; with
// the few micro-batches a pipeline keeps in flight it never has to). A token more than kSlots
// transfers old names a transfer that has therefore completed: waiting for it is a no-op.
int slot_of(int token) { return (int)((uint32_t)token % kSlots); }

bool token_expired(const Edge* e, int token) {
  return ((e->next - (uint32_t)token) & kTokenMask) > (uint32_t)kSlots;
}

int take_token(Edge* e) {
  const uint32_t t = e->next & kTokenMask;
  const int s = (int)(t % kSlots);
  if (e->used[s]) {
    auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      hipError_t q = hipEventQuery(e->done[s]);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) return fail("hipEventQuery", hipGetErrorString(q), (int)q);
      if (e->aborted.load()) return fail("completion ring", "edge aborted", 0);
      if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > kRingWaitMs)
        return fail("completion ring", "256 transfers in flight and the oldest does not complete", 0);
      e->ring_waits++;
      std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
  }
  e->used[s] = true;
  e->next = (e->next + 1) & kTokenMask;
  return (int)t;
}

// `after_stream` is a stream handle, where null IS a stream (the default one); "no stream" is
// DLIA_NO_STREAM.
int fence_from(Edge* e, int slot, void* after_stream, hipEvent_t after_event) {
  if (after_stream != DLIA_NO_STREAM) {
    HIPCHK(hipEventRecord(e->fence[slot], (hipStream_t)after_stream));
    HIPCHK(hipStreamWaitEvent(e->stream, e->fence[slot], 0));
  }
  if (after_event) HIPCHK(hipStreamWaitEvent(e->stream, after_event, 0));
  return 0;
}

}  // namespace

extern "C" {

const char* dlia_comm_last_error() { return t_err; }

int dlia_comm_init(const char* rccl_path) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_api.lib) return 0;
  void* h = dlopen(rccl_path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail("dlopen", dlerror(), 0);
#define SYM(field, name)                                                   \
  *(void**)(&g_api.field) = dlsym(h, name);                                \
  if (!g_api.field) { return fail("dlsym", name, 0); }
  SYM(GetVersion, "ncclGetVersion")
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(CommAbort, "ncclCommAbort")
  SYM(CommGetAsyncError, "ncclCommGetAsyncError")
  SYM(GetErrorString, "ncclGetErrorString")
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
  g_api.lib = h;
  return 0;
}

int dlia_comm_version() {
  int v = 0;
  if (!g_api.lib || g_api.GetVersion(&v) != 0) return -1;
  return v;
}

// 128 bytes that every rank of one communicator must pass to dlia_edge_open.
int dlia_comm_unique_id(void* out128) {
  if (!g_api.lib) return fail("dlia_comm_unique_id", "dlia_comm_init not called", 0);
  ncclUniqueId id;
  NCCLCHK(g_api.GetUniqueId(&id));
  memcpy(out128, id.internal, 128);
  return 0;
}

// Collective over the `nranks` ranks holding `uid`: blocks until all of them have called it.
int dlia_edge_open(const void* uid128, int nranks, int rank, int peer, int device, void** out) {
  if (!g_api.lib) return fail("dlia_edge_open", "dlia_comm_init not called", 0);
  if (nranks < 1 || rank < 0 || rank >= nranks || peer < 0 || peer >= nranks)
    return fail("dlia_edge_open", "rank / peer out of range", 0);
  HIPCHK(hipSetDevice(device));
  Edge* e = new (std::nothrow) Edge();
  if (!e) return fail("dlia_edge_open", "out of memory", 0);
  e->nranks = nranks; e->rank = rank; e->peer = peer; e->device = device;
  // Default priority on purpose: on ROCm 7.0 / gfx950 a high-priority stream that waits for an
  // event of a default-priority stream still busy (the fence of a send behind long kernels) never
  // resumes when both share one of the 4 default hardware queues — the hop hung the device.
  HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  for (int i = 0; i < kSlots; ++i) {
    HIPCHK(hipEventCreateWithFlags(&e->done[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->fence[i], hipEventDisableTiming));
  }
  ncclUniqueId id;
  memcpy(id.internal, uid128, 128);
  NCCLCHK(g_api.CommInitRank(&e->comm, nranks, id, rank));
  *out = e;
  return 0;
}

// ---- asynchronous hop on the edge's own stream ------------------------------------------------
// `after_stream`: work already enqueued there must finish before the transfer touches the buffer
// (the producer of a send; the previous reader of a receive buffer); DLIA_NO_STREAM for none.
// Returns a completion token.
int dlia_edge_send(void* h, const void* ptr, size_t nbytes, void* after_stream) {
  Edge* e = static_cast<Edge*>(h);
  if (e->aborted.load()) return fail("dlia_edge_send", "edge aborted", 0);
  DeviceScope ds(e->device);
  const int token = take_token(e);
  if (token < 0) return -1;
  const int slot = slot_of(token);
  if (fence_from(e, slot, after_stream, nullptr)) return -1;
  NCCLCHK(g_api.Send(ptr, nbytes, kNcclInt8, e->peer, e->comm, e->stream));
  HIPCHK(hipEventRecord(e->done[slot], e->stream));
  e->bytes_sent += nbytes; e->ops++;
  return token;
}

int dlia_edge_recv(void* h, void* ptr, size_t nbytes, void* after_stream, void* after_event) {
  Edge* e = static_cast<Edge*>(h);
  if (e->aborted.load()) return fail("dlia_edge_recv", "edge aborted", 0);
  DeviceScope ds(e->device);
  const int token = take_token(e);
  if (token < 0) return -1;
  const int slot = slot_of(token);
  if (fence_from(e, slot, after_stream, (hipEvent_t)after_event)) return -1;
  NCCLCHK(g_api.Recv(ptr, nbytes, kNcclInt8, e->peer, e->comm, e->stream));
  HIPCHK(hipEventRecord(e->done[slot], e->stream));
  e->bytes_recv += nbytes; e->ops++;
  return token;
}

// Send and receive with the peer as one group (a rank exchanging with itself on a one-rank
// communicator is the degenerate case: the only way to run the wire path on a single GPU).
int dlia_edge_sendrecv(void* h, const void* sptr, void* rptr, size_t nbytes, void* after_stream) {
  Edge* e = static_cast<Edge*>(h);
  if (e->aborted.load()) return fail("dlia_edge_sendrecv", "edge aborted", 0);
  DeviceScope ds(e->device);
  const int token = take_token(e);
  if (token < 0) return -1;
  const int slot = slot_of(token);
  if (fence_from(e, slot, after_stream, nullptr)) return -1;
  NCCLCHK(g_api.GroupStart());
  ncclResult_t r1 = g_api.Send(sptr, nbytes, kNcclInt8, e->peer, e->comm, e->stream);
  ncclResult_t r2 = g_api.Recv(rptr, nbytes, kNcclInt8, e->peer, e->comm, e->stream);
  ncclResult_t r3 = g_api.GroupEnd();
  if (r1 || r2 || r3) return fail("grouped send/recv", g_api.GetErrorString(r1 ? r1 : (r2 ? r2 : r3)), r1 ? r1 : (r2 ? r2 : r3));
  HIPCHK(hipEventRecord(e->done[slot], e->stream));
  e->bytes_sent += nbytes; e->bytes_recv += nbytes; e->ops++;
  return token;
}

// ---- the same transfers on the caller's stream (capturable into a hipGraph) --------------------
int dlia_edge_send_on(void* h, const void* ptr, size_t nbytes, void* stream) {
  Edge* e = static_cast<Edge*>(h);
  if (e->aborted.load()) return fail("dlia_edge_send_on", "edge aborted", 0);
  DeviceScope ds(e->device);
  NCCLCHK(g_api.Send(ptr, nbytes, kNcclInt8, e->peer, e->comm, (hipStream_t)stream));
  return 0;
}

int dlia_edge_recv_on(void* h, void* ptr, size_t nbytes, void* stream) {
  Edge* e = static_cast<Edge*>(h);
  if (e->aborted.load()) return fail("dlia_edge_recv_on", "edge aborted", 0);
  DeviceScope ds(e->device);
  NCCLCHK(g_api.Recv(ptr, nbytes, kNcclInt8, e->peer, e->comm, (hipStream_t)stream));
  return 0;
}

int dlia_edge_sendrecv_on(void* h, const void* sptr, void* rptr, size_t nbytes, void* stream) {
  Edge* e = static_cast<Edge*>(h);
  if (e->aborted.load()) return fail("dlia_edge_sendrecv_on", "edge aborted", 0);
  DeviceScope ds(e->device);
  NCCLCHK(g_api.GroupStart());
  ncclResult_t r1 = g_api.Send(sptr, nbytes, kNcclInt8, e->peer, e->comm, (hipStream_t)stream);
  ncclResult_t r2 = g_api.Recv(rptr, nbytes, kNcclInt8, e->peer, e->comm, (hipStream_t)stream);
  ncclResult_t r3 = g_api.GroupEnd();
  if (r1 || r2 || r3) return fail("grouped send/recv", g_api.GetErrorString(r1 ? r1 : (r2 ? r2 : r3)), r1 ? r1 : (r2 ? r2 : r3));
  return 0;
}

// ---- completion ----------------------------------------------------------------------------------
// Make `stream` wait for the transfer behind `token` (no host blocking).
int dlia_edge_wait(void* h, int token, void* stream) {
  Edge* e = static_cast<Edge*>(h);
  if (token < 0 || (uint32_t)token > kTokenMask) return fail("dlia_edge_wait", "bad token", token);
  if (token_expired(e, token)) return 0;              // completed long ago (its slot has been reused since)
  HIPCHK(hipStreamWaitEvent((hipStream_t)stream, e->done[slot_of(token)], 0));
  return 0;
}

// 1 = complete, 0 = still in flight, -1 = the communicator reported an asynchronous error.
int dlia_edge_query(void* h, int token) {
  Edge* e = static_cast<Edge*>(h);
  if (token < 0 || (uint32_t)token > kTokenMask) return fail("dlia_edge_query", "bad token", token);
  if (token_expired(e, token)) return 1;
  hipError_t q = hipEventQuery(e->done[slot_of(token)]);
  if (q == hipSuccess) return 1;
  if (q != hipErrorNotReady) return fail("hipEventQuery", hipGetErrorString(q), (int)q);
  if (e->aborted.load()) return fail("dlia_edge_query", "edge aborted", 0);
  ncclResult_t async = 0;
  if (g_api.CommGetAsyncError(e->comm, &async) == 0 && async != 0 && async != 7)
    return fail("rccl async error", g_api.GetErrorString(async), async);
  return 0;
}

// Host-side wait with a deadline: 0 = complete, 1 = timed out, -1 = error. Polls, never parks in
// the driver, so a peer that died leaves the caller able to abort the edge.
int dlia_edge_sync(void* h, int token, int timeout_ms) {
  auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  for (;;) {
    int q = dlia_edge_query(h, token);
    if (q != 0) return q == 1 ? 0 : -1;
    auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (timeout_ms >= 0 && ms >= timeout_ms) return 1;
    if (++spins < 200) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

void* dlia_edge_stream(void* h) { return static_cast<Edge*>(h)->stream; }

void dlia_edge_stats(void* h, uint64_t* out4) {
  Edge* e = static_cast<Edge*>(h);
  out4[0] = e->bytes_sent; out4[1] = e->bytes_recv; out4[2] = e->ops; out4[3] = e->ring_waits;
}

// Tear the communicator down without waiting for its outstanding transfers: kernels spinning on a
// peer that will never answer are released, queued work on the edge's stream is abandoned. After
// this only dlia_edge_close is valid.
int dlia_edge_abort(void* h) {
  Edge* e = static_cast<Edge*>(h);
  if (e->aborted.exchange(1)) return 0;
  DeviceScope ds(e->device);
  if (e->comm) { NCCLCHK(g_api.CommAbort(e->comm)); e->comm = nullptr; }
  return 0;
}

// Destroys the edge. A stream that still has transfers in flight is aborted rather than drained:
// closing must never hang on a dead peer.
int dlia_edge_close(void* h) {
  Edge* e = static_cast<Edge*>(h);
  DeviceScope ds(e->device);
  if (e->comm && !e->aborted.load()) {
    if (hipStreamQuery(e->stream) == hipSuccess) g_api.CommDestroy(e->comm);
    else g_api.CommAbort(e->comm);
    e->comm = nullptr;
  }
  for (int i = 0; i < kSlots; ++i) { hipEventDestroy(e->done[i]); hipEventDestroy(e->fence[i]); }
  if (e->stream) hipStreamDestroy(e->stream);
  delete e;
  return 0;
}

}  // extern "C"
