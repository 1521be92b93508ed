// One directed pipeline hop between two stages that live in ONE process on one GPU (stage threads of
// an in-process pipeline): same C ABI as the RCCL hop (rccl_edge.h), so the native stage loop and
// the Python wrappers drive either. A send copies the buffer into a slot of a device staging ring
// on the edge's send stream, behind the producer's work; a receive copies it out on the edge's
// receive stream, behind the buffer's previous reader; the consumer stream waits for the receive's
// completion event. Nothing blocks the host except a receive posted before its send has been
// ENQUEUED (it then waits, with a deadline, for the sender thread to get there).
//
// The reference's hop is an HTTP POST of the whole hidden-state sequence as JSON
// (orchestration.py:114-137; Worker1.py:208-245).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <new>
#include <thread>

#include "rccl_edge.h"

namespace {

constexpr int kRing = 8;               // staged messages in flight
constexpr int kTokens = 256;           // completion events in flight per direction

thread_local char t_lerr[256];
int lfail(const char* what, hipError_t e) {
  snprintf(t_lerr, sizeof(t_lerr), "%s: %s", what, hipGetErrorString(e));
  return -1;
}
#define LCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return lfail(#expr, e_); } while (0)

struct LocalEdge {
  int device = 0;
  size_t cap = 0;                      // bytes per staging slot
  char* staging = nullptr;             // kRing x cap
  size_t bytes[kRing] = {};
  hipStream_t sstream = nullptr, rstream = nullptr;
  hipEvent_t filled[kRing], drained[kRing];
  hipEvent_t sdone[kTokens], rdone[kTokens], sfence[kTokens], rfence[kTokens];
  std::atomic<uint64_t> sent{0}, received{0};   // messages whose copy-in / copy-out has been ENQUEUED
  std::atomic<int> aborted{0};
  int recv_timeout_ms = 60000;
  uint64_t bytes_sent = 0, bytes_recv = 0;
};

struct DevScope {
  int prev = -1; bool changed = false;
  explicit DevScope(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess); }
  ~DevScope() { if (changed) hipSetDevice(prev); }
};

}  // namespace

extern "C" {

const char* dlia_ledge_last_error() { return t_lerr; }

// `max_bytes`: the largest message (a prefill chunk's hidden states).
int dlia_ledge_open(int device, size_t max_bytes, void** out) {
  DevScope ds(device);
  LocalEdge* e = new (std::nothrow) LocalEdge();
  if (!e) { snprintf(t_lerr, sizeof(t_lerr), "out of memory"); return -1; }
  e->device = device; e->cap = (max_bytes + 255) & ~(size_t)255;
  LCHK(hipMalloc((void**)&e->staging, e->cap * kRing));
  LCHK(hipStreamCreateWithFlags(&e->sstream, hipStreamNonBlocking));
  LCHK(hipStreamCreateWithFlags(&e->rstream, hipStreamNonBlocking));
  for (int i = 0; i < kRing; ++i) {
    LCHK(hipEventCreateWithFlags(&e->filled[i], hipEventDisableTiming));
    LCHK(hipEventCreateWithFlags(&e->drained[i], hipEventDisableTiming));
  }
  for (int i = 0; i < kTokens; ++i) {
    LCHK(hipEventCreateWithFlags(&e->sdone[i], hipEventDisableTiming));
    LCHK(hipEventCreateWithFlags(&e->rdone[i], hipEventDisableTiming));
    LCHK(hipEventCreateWithFlags(&e->sfence[i], hipEventDisableTiming));
    LCHK(hipEventCreateWithFlags(&e->rfence[i], hipEventDisableTiming));
  }
  *out = e;
  return 0;
}

// Sender thread only. Token: completion of the copy out of `ptr` (the buffer may be reused after it).
int dlia_ledge_send(void* h, const void* ptr, size_t nbytes, void* after_stream) {
  LocalEdge* e = static_cast<LocalEdge*>(h);
  if (e->aborted.load()) { snprintf(t_lerr, sizeof(t_lerr), "edge aborted"); return -1; }
  if (nbytes > e->cap) { snprintf(t_lerr, sizeof(t_lerr), "message of %zu bytes exceeds the edge's %zu", nbytes, e->cap); return -1; }
  DevScope ds(e->device);
  const uint64_t i = e->sent.load(std::memory_order_relaxed);
  const int slot = (int)(i % kRing), tok = (int)(i % kTokens);
  // the slot's previous message must have been copied out: its receive is at least ENQUEUED once
  // `received` has passed it, and then its `drained` event orders the overwrite on the device
  auto t0 = std::chrono::steady_clock::now();
  while (i >= kRing && e->received.load(std::memory_order_acquire) + kRing <= i) {
    if (e->aborted.load()) { snprintf(t_lerr, sizeof(t_lerr), "edge aborted"); return -1; }
    if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > e->recv_timeout_ms) {
      snprintf(t_lerr, sizeof(t_lerr), "local edge: the receiver is %d messages behind", kRing);
      return -1;
    }
    std::this_thread::yield();
  }
  if (i >= kRing) LCHK(hipStreamWaitEvent(e->sstream, e->drained[slot], 0));
  if (after_stream != DLIA_NO_STREAM) {
    LCHK(hipEventRecord(e->sfence[tok], (hipStream_t)after_stream));
    LCHK(hipStreamWaitEvent(e->sstream, e->sfence[tok], 0));
  }
  if (nbytes) LCHK(hipMemcpyAsync(e->staging + (size_t)slot * e->cap, ptr, nbytes, hipMemcpyDeviceToDevice, e->sstream));
  LCHK(hipEventRecord(e->filled[slot], e->sstream));
  LCHK(hipEventRecord(e->sdone[tok], e->sstream));
  e->bytes[slot] = nbytes;
  e->bytes_sent += nbytes;
  e->sent.store(i + 1, std::memory_order_release);
  return tok;
}

// Receiver thread only. Token: completion of the copy into `ptr`.
int dlia_ledge_recv(void* h, void* ptr, size_t nbytes, void* after_stream, void* after_event) {
  LocalEdge* e = static_cast<LocalEdge*>(h);
  DevScope ds(e->device);
  const uint64_t j = e->received.load(std::memory_order_relaxed);
  const int slot = (int)(j % kRing), tok = (int)(j % kTokens);
  auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (e->sent.load(std::memory_order_acquire) <= j) {       // the matching send has not been enqueued yet
    if (e->aborted.load()) { snprintf(t_lerr, sizeof(t_lerr), "edge aborted"); return -1; }
    if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > e->recv_timeout_ms) {
      snprintf(t_lerr, sizeof(t_lerr), "local edge: no message within %d ms", e->recv_timeout_ms);
      return -1;
    }
    if (++spins < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  if (e->bytes[slot] != nbytes) {
    snprintf(t_lerr, sizeof(t_lerr), "local edge: expected %zu bytes, got %zu", nbytes, e->bytes[slot]);
    return -1;
  }
  if (after_stream != DLIA_NO_STREAM) {
    LCHK(hipEventRecord(e->rfence[tok], (hipStream_t)after_stream));
    LCHK(hipStreamWaitEvent(e->rstream, e->rfence[tok], 0));
  }
  if (after_event) LCHK(hipStreamWaitEvent(e->rstream, (hipEvent_t)after_event, 0));
  LCHK(hipStreamWaitEvent(e->rstream, e->filled[slot], 0));
  if (nbytes) LCHK(hipMemcpyAsync(ptr, e->staging + (size_t)slot * e->cap, nbytes, hipMemcpyDeviceToDevice, e->rstream));
  LCHK(hipEventRecord(e->drained[slot], e->rstream));
  LCHK(hipEventRecord(e->rdone[tok], e->rstream));
  e->bytes_recv += nbytes;
  e->received.store(j + 1, std::memory_order_release);
  return tok;
}

// `recv_side`: which end's token this is (the two ends count their tokens separately).
static hipEvent_t ledge_event(LocalEdge* e, int token, int recv_side) { return recv_side ? e->rdone[token] : e->sdone[token]; }

int dlia_ledge_wait(void* h, int token, int recv_side, void* stream) {
  LocalEdge* e = static_cast<LocalEdge*>(h);
  if (token < 0 || token >= kTokens) { snprintf(t_lerr, sizeof(t_lerr), "bad token"); return -1; }
  LCHK(hipStreamWaitEvent((hipStream_t)stream, ledge_event(e, token, recv_side), 0));
  return 0;
}
// the stage loop's view: one function per end, same signature as dlia_edge_wait
// The stream sends run on (for timing events recorded behind a send).
void* dlia_ledge_send_stream(void* h) { return static_cast<LocalEdge*>(h)->sstream; }

int dlia_ledge_wait_send(void* h, int token, void* stream) { return dlia_ledge_wait(h, token, 0, stream); }
int dlia_ledge_wait_recv(void* h, int token, void* stream) { return dlia_ledge_wait(h, token, 1, stream); }

// 1 complete, 0 in flight, -1 error
int dlia_ledge_query(void* h, int token, int recv_side) {
  LocalEdge* e = static_cast<LocalEdge*>(h);
  if (token < 0 || token >= kTokens) { snprintf(t_lerr, sizeof(t_lerr), "bad token"); return -1; }
  hipError_t q = hipEventQuery(ledge_event(e, token, recv_side));
  if (q == hipSuccess) return 1;
  if (q == hipErrorNotReady) return e->aborted.load() ? -1 : 0;
  return lfail("hipEventQuery", q);
}

// Messages whose send has been enqueued and whose receive has not been posted yet: a receive posted while this
// is > 0 does not wait on the host.
int dlia_ledge_pending(void* h) {
  LocalEdge* e = static_cast<LocalEdge*>(h);
  const uint64_t s = e->sent.load(std::memory_order_acquire), r = e->received.load(std::memory_order_acquire);
  return s > r ? (int)(s - r) : 0;
}

void dlia_ledge_set_timeout(void* h, int ms) { static_cast<LocalEdge*>(h)->recv_timeout_ms = ms; }

int dlia_ledge_abort(void* h) { static_cast<LocalEdge*>(h)->aborted.store(1); return 0; }

int dlia_ledge_close(void* h) {
  LocalEdge* e = static_cast<LocalEdge*>(h);
  DevScope ds(e->device);
  (void)hipStreamSynchronize(e->sstream);
  (void)hipStreamSynchronize(e->rstream);
  for (int i = 0; i < kRing; ++i) { hipEventDestroy(e->filled[i]); hipEventDestroy(e->drained[i]); }
  for (int i = 0; i < kTokens; ++i) { hipEventDestroy(e->sdone[i]); hipEventDestroy(e->rdone[i]); hipEventDestroy(e->sfence[i]); hipEventDestroy(e->rfence[i]); }
  if (e->staging) (void)hipFree(e->staging);
  hipStreamDestroy(e->sstream); hipStreamDestroy(e->rstream);
  delete e;
  return 0;
}

}  // extern "C"
