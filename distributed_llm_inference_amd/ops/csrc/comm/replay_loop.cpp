// Native serving loop of a single-stage pipeline's stage process.
//
// In steady state a decode step of a micro-batch is "the same captured graph once more": the
// scheduler sends a 40-byte compact replay descriptor (runtime/batch.py) and wants the sampled token
// ids back. This loop does that without entering Python: descriptor ring -> (block-table growth
// entries: 4-byte host-to-device copies) -> hipGraphLaunch on the micro-batch's stream -> copy of the
// token ids to pinned host memory on the same stream -> completion event -> result ring. Everything
// else — prefill chunks, recompositions, control messages — is handed back to the Python loop
// (runtime/head.py::head_main), which answers through the same queue so that the result ring keeps
// its single producer; results leave as their steps complete.
//
// The same loop serves a stage behind stage 0 of a multi-stage pipeline (dlia_stage_*): a replay
// descriptor from the previous stage's ring -> block-table growth -> receive of the hidden states
// into the graph's input buffer (the hop's own stream; the micro-batch's stream waits for it) ->
// hipGraphLaunch -> descriptor on to the next stage's ring -> send of the output buffer (the last
// stage: the sampled ids back to stage 0). The hop is reached through a table of three functions,
// so the RCCL hop between processes (rccl_edge.cpp) and the in-process hop (local_edge.cpp) both fit.
//
// And stage 0 of a multi-stage pipeline (dlia_head_serve): descriptor from the scheduler's ring -> growth ->
// wait for the micro-batch's fed-back ids (the receive the previous step posted into the graph's token
// buffer) -> hipGraphLaunch -> descriptor on -> send of the hidden states -> receive of THIS step's ids
// from the last stage posted behind the graph -> their copy to pinned host memory on a side stream ->
// result ring. `in` is then the token feedback hop, `out` the hop to stage 1.
//
// The reference's counterpart is the Flask request handler around model.forward
// (Worker1.py:208-245; orchestration.py:109-196).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <new>
#include <thread>
#include <vector>

#include "rccl_edge.h"

namespace {

constexpr uint32_t kDescMagic = 0x444C4942;     // "DLIB" (runtime/batch.py)
constexpr uint32_t kResMagic = 0x52455331;      // "RES1" (runtime/head.py)
constexpr int kKindStep = 1;
constexpr uint32_t kFlagReplay = 1, kFlagCompact = 2;
constexpr int kSlots = 8;                        // results in flight per micro-batch
constexpr int kMaxGrow = 64;                     // growth entries handled here per step; more go to Python

struct DescHead {                                // "<IiiiiiiIii"
  uint32_t magic; int32_t kind, step_id, mb, rows, ntok, ntb; uint32_t flags; int32_t ngrow, live;
};
static_assert(sizeof(DescHead) == 40, "descriptor header layout");

typedef int (*ring_get_fn)(void*, void*, uint32_t, int);
typedef int (*ring_put_fn)(void*, const void*, uint32_t, int);

typedef int (*edge_send_fn)(void*, const void*, size_t, void*);
typedef int (*edge_recv_fn)(void*, void*, size_t, void*, void*);
typedef int (*edge_wait_fn)(void*, int, void*);
typedef const char* (*edge_err_fn)();
struct EdgeRef {                                 // one end of a hop: the receiving end of `in`, the sending end of `out`
  void* h = nullptr; edge_send_fn send = nullptr; edge_recv_fn recv = nullptr; edge_wait_fn wait = nullptr;
  edge_err_fn err = nullptr;                     // the hop library's last error text (this thread's)
};

struct Mb {
  hipGraphExec_t exec = nullptr;
  hipStream_t stream = nullptr;
  const int32_t* tokens = nullptr;               // device: the graph's sampled ids
  int rows = 0;
  // a stage behind stage 0: the graph's input (hidden states) and output (hidden states, or the ids on the last stage)
  void* in_ptr = nullptr; size_t in_bytes = 0;
  const void* out_ptr = nullptr; size_t out_bytes = 0;
  int send_tok = -1;                             // completion of the latest send out of out_ptr
  int fb_tok = -1;                               // stage 0 of several: arrival of the ids the next step reads (in_ptr)
  int last_slot = -1;                            //   and the host slot they were last copied to (its event guards in_ptr)
  // a stage behind stage 0: end of the graph launch that last read in_ptr, recorded by THIS loop. While it is valid
  // the next receive into in_ptr waits for that event only — not for everything queued on the micro-batch's stream,
  // which several micro-batches share: the hidden states of micro-batch k + 2 then arrive while micro-batch k still
  // computes on that stream, and the sender's hop stream is not held up behind this stage's compute.
  hipEvent_t in_free = nullptr; bool in_free_ok = false;
  int32_t* host[kSlots] = {};
  hipEvent_t ev[kSlots] = {};
  int next = 0;
};

// ---- step timeline -----------------------------------------------------------------------------------
// Every step this loop issues is bracketed by HIP timing events on the streams it runs on: when the
// micro-batch's stream became free for it (e[0]), when its input had arrived and the graph started
// (e[1]), when the graph ended (e[2]) and, on a stage that sends, when the send had left (e[3], on the
// hop's stream). Completed steps are turned into spans on the host clock (CLOCK_MONOTONIC, the clock
// Python's perf_counter reads): recv-wait = e[0]..e[1], compute = e[1]..e[2], send = e[2]..e[3], next to
// the host time this loop spent enqueueing the step. The compute spans also give the stage's GPU-busy
// time (union over overlapping micro-batch streams), which is what /metrics reports per stage. Python
// reads the spans out for a Chrome trace; nobody has to: the ring of finished spans overwrites itself.
constexpr int kOpenSteps = 256;                  // steps whose events have not all completed yet
constexpr int kDoneSpans = 8192;                 // finished spans kept for read-out
constexpr int kSpanDoubles = 9;                  // step id, micro-batch, rows, host0, host1, recv0, comp0, comp1, send1 (microseconds)

struct StepRec {
  int step_id = 0, mb = 0, rows = 0, anchor = 0;
  int64_t host0_ns = 0, host1_ns = 0;
  hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
};

struct Anchor { hipEvent_t ev = nullptr; int64_t host_ns = 0; };

struct Timeline {
  bool on = true;
  Anchor anchors[2]; int cur = 0; int64_t anchored_ns = 0;
  std::vector<hipEvent_t> pool;
  std::deque<StepRec> open;
  std::vector<double> done; size_t done_head = 0, done_count = 0;   // ring of kDoneSpans x kSpanDoubles
  uint64_t busy_ns = 0, recv_wait_ns = 0, send_ns = 0, dropped = 0, tokens = 0;
  double busy_until_us = 0;
};

inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Pending {                                 // a submitted step whose result has not left yet
  int step_id; int n; const int32_t* host; hipEvent_t ev;
};

struct Loop {
  int device = 0;
  std::vector<Mb> mbs;
  std::deque<Pending> pending;
  ring_get_fn rget = nullptr; ring_put_fn rput = nullptr;
  int32_t* table = nullptr; int table_rows = 0, max_blocks = 0, num_blocks = 0;   // device block table
  int32_t* grow_stage = nullptr; int grow_pos = 0;                                  // pinned staging for growth values
  std::vector<uint8_t> out;
  uint64_t native_steps = 0, handed_up = 0;
  EdgeRef ein, eout;                             // stage loop only: the hop in, the hop out
  hipStream_t out_stream = nullptr;              // stream the outgoing hop's transfers run on (null: unknown, no send spans)
  hipStream_t res_stream = nullptr;              // stage 0 of several: where fed-back ids are waited for and copied to the host
  int64_t handoff_in = -1;                       // > 0: hand the n-th replay step from now up to Python instead of serving it
  Timeline tl;
  char err[256] = {0};
};

int fail(Loop* l, const char* what, hipError_t e) {
  snprintf(l->err, sizeof(l->err), "%s: %s", what, hipGetErrorString(e));
  return -1;
}
#define HIPCHK(l, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(l, #expr, e_); } while (0)

// A device-clock anchor: an event stamped with the host clock at its completion. It is recorded on the stream of the
// step being opened and waited for — NOT on a stream of its own: one more stream in the process made two micro-batch
// streams share a hardware queue (HIP maps streams onto GPU_MAX_HW_QUEUES queues, 4 by default) and cost 9-13 % of
// the throughput of 256-row steps, while the timing events themselves cost nothing measurable. The wait is for
// whatever that stream has queued: nothing at the first step, a few steps when the anchor is renewed (every few
// minutes, so that event-to-event distances stay small numbers in float milliseconds; the previous anchor stays
// valid for the steps that were opened under it).
int tl_anchor(Loop* l, hipStream_t stream) {
  Timeline& t = l->tl;
  const int64_t n = now_ns();
  if (t.anchors[t.cur].ev && n - t.anchored_ns < 300000000000LL) return 0;
  const int nxt = t.anchors[t.cur].ev ? 1 - t.cur : t.cur;
  Anchor& a = t.anchors[nxt];
  if (!a.ev && hipEventCreate(&a.ev) != hipSuccess) return -1;
  if (hipEventRecord(a.ev, stream) != hipSuccess || hipEventSynchronize(a.ev) != hipSuccess) return -1;
  a.host_ns = now_ns();
  t.cur = nxt; t.anchored_ns = a.host_ns;
  return 0;
}

hipEvent_t tl_event(Timeline& t) {
  if (!t.pool.empty()) { hipEvent_t e = t.pool.back(); t.pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

// Opens the record of a step about to be issued; null when the timeline is off or too many steps are open.
StepRec* tl_open(Loop* l, int step_id, int mb, int rows, int live, hipStream_t stream) {
  Timeline& t = l->tl;
  t.tokens += (uint64_t)(live > 0 ? live : 0);
  if (!t.on) return nullptr;
  if ((int)t.open.size() >= kOpenSteps || tl_anchor(l, stream) != 0) { t.dropped++; return nullptr; }
  t.open.emplace_back();
  StepRec& r = t.open.back();
  r.step_id = step_id; r.mb = mb; r.rows = rows; r.anchor = t.cur; r.host0_ns = now_ns();
  return &r;
}

void tl_stamp(Loop* l, StepRec* r, int which, hipStream_t stream) {
  if (!r) return;
  hipEvent_t e = tl_event(l->tl);
  if (e && hipEventRecord(e, stream) == hipSuccess) r->e[which] = e;
  else if (e) l->tl.pool.push_back(e);
}

// Turns the steps whose events have all completed into spans; cheap when nothing is open.
void tl_reap(Loop* l) {
  Timeline& t = l->tl;
  while (!t.open.empty()) {
    StepRec& r = t.open.front();
    bool ready = true;
    for (int i = 3; i >= 0 && ready; --i)
      if (r.e[i] && hipEventQuery(r.e[i]) == hipErrorNotReady) ready = false;
    if (!ready) return;
    const Anchor& a = t.anchors[r.anchor];
    double at[4] = {-1, -1, -1, -1};
    for (int i = 0; i < 4; ++i) {
      float ms = 0.f;
      if (r.e[i] && a.ev && hipEventElapsedTime(&ms, a.ev, r.e[i]) == hipSuccess) at[i] = (double)a.host_ns / 1e3 + (double)ms * 1e3;
      if (r.e[i]) t.pool.push_back(r.e[i]);
    }
    if (at[1] >= 0 && at[2] >= at[1]) {            // GPU-busy time: union of the compute spans of all micro-batch streams
      const double b = at[1] > t.busy_until_us ? at[1] : t.busy_until_us;
      if (at[2] > b) { t.busy_ns += (uint64_t)((at[2] - b) * 1e3); t.busy_until_us = at[2]; }
    }
    if (at[0] >= 0 && at[1] >= at[0]) t.recv_wait_ns += (uint64_t)((at[1] - at[0]) * 1e3);
    if (at[2] >= 0 && at[3] >= at[2]) t.send_ns += (uint64_t)((at[3] - at[2]) * 1e3);
    if (t.done.empty()) t.done.resize((size_t)kDoneSpans * kSpanDoubles);
    const size_t slot = (t.done_head + t.done_count) % kDoneSpans;
    if (t.done_count == (size_t)kDoneSpans) { t.done_head = (t.done_head + 1) % kDoneSpans; t.dropped++; } else t.done_count++;
    double* o = &t.done[slot * kSpanDoubles];
    o[0] = r.step_id; o[1] = r.mb; o[2] = r.rows; o[3] = (double)r.host0_ns / 1e3; o[4] = (double)r.host1_ns / 1e3;
    o[5] = at[0]; o[6] = at[1]; o[7] = at[2]; o[8] = at[3];
    t.open.pop_front();
  }
}

// Sends every queued result whose event has completed. Results carry their step id and the
// scheduler matches them by it, so they leave as they complete, not in submission order: the tokens
// of a decode step must not wait behind a prefill chunk of another micro-batch that was submitted
// earlier and runs for longer (steps of ONE micro-batch share a stream and complete in order
// anyway). 0 ok, -1 error.
int drain(Loop* l, void* res_ring) {
  for (auto it = l->pending.begin(); it != l->pending.end();) {
    Pending& p = *it;
    if (p.ev) {
      hipError_t q = hipEventQuery(p.ev);
      if (q == hipErrorNotReady) { ++it; continue; }
      if (q != hipSuccess) return fail(l, "hipEventQuery", q);
    }
    const size_t bytes = 16 + (size_t)p.n * 4;
    if (l->out.size() < bytes) l->out.resize(bytes);
    uint32_t head[4] = {kResMagic, (uint32_t)p.step_id, (uint32_t)p.n, 0u};
    memcpy(l->out.data(), head, 16);
    if (p.n) memcpy(l->out.data() + 16, p.host, (size_t)p.n * 4);
    if (l->rput(res_ring, l->out.data(), (uint32_t)bytes, 5000) != 0) {
      snprintf(l->err, sizeof(l->err), "result ring: put failed");
      return -1;
    }
    it = l->pending.erase(it);
  }
  return 0;
}

}  // namespace

extern "C" {

void* dlia_replay_create(int device, int max_mbs, void* ring_get, void* ring_put) {
  Loop* l = new (std::nothrow) Loop();
  if (!l) return nullptr;
  l->device = device;
  l->mbs.resize(max_mbs > 0 ? max_mbs : 1);
  l->rget = (ring_get_fn)ring_get; l->rput = (ring_put_fn)ring_put;
  if (hipSetDevice(device) != hipSuccess ||
      hipHostMalloc((void**)&l->grow_stage, 4096 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
    delete l;
    return nullptr;
  }
  return l;
}

const char* dlia_replay_error(void* h) { return static_cast<Loop*>(h)->err; }

// The stage's block table (device, int32 [table_rows][max_blocks]) for growth entries.
int dlia_replay_set_table(void* h, void* table, int table_rows, int max_blocks, int num_blocks) {
  Loop* l = static_cast<Loop*>(h);
  l->table = (int32_t*)table; l->table_rows = table_rows; l->max_blocks = max_blocks; l->num_blocks = num_blocks;
  return 0;
}

// Registers (or replaces) the captured decode step of micro-batch `mb`: its executable graph, the
// stream it runs on, where its sampled ids land and how many there are. exec == null unregisters.
int dlia_replay_set_mb(void* h, int mb, void* exec, void* stream, const void* tokens, int rows) {
  Loop* l = static_cast<Loop*>(h);
  if (mb < 0 || mb >= (int)l->mbs.size()) { snprintf(l->err, sizeof(l->err), "micro-batch %d out of range", mb); return -1; }
  Mb& m = l->mbs[mb];
  if (m.rows != rows) {
    for (int i = 0; i < kSlots; ++i) {
      if (m.host[i]) { (void)hipHostFree(m.host[i]); m.host[i] = nullptr; }
      if (!m.ev[i]) HIPCHK(l, hipEventCreateWithFlags(&m.ev[i], hipEventDisableTiming));
      if (rows > 0) HIPCHK(l, hipHostMalloc((void**)&m.host[i], (size_t)rows * 4, hipHostMallocDefault));
    }
  }
  m.exec = (hipGraphExec_t)exec; m.stream = (hipStream_t)stream; m.tokens = (const int32_t*)tokens; m.rows = rows;
  return 0;
}

// A step the Python loop ran itself: its result joins the queue behind the native ones. `event`
// (a HIP event handle, may be null = already complete) guards `host` (n ids, stays valid until sent).
int dlia_replay_push_external(void* h, int step_id, void* event, const void* host, int n) {
  Loop* l = static_cast<Loop*>(h);
  l->pending.push_back(Pending{step_id, n, (const int32_t*)host, (hipEvent_t)event});
  return 0;
}

int dlia_replay_pending(void* h) { return (int)static_cast<Loop*>(h)->pending.size(); }

// out8: steps served here, messages handed to Python, GPU-busy ns (union of compute spans), ns the
// micro-batch streams waited for their input, ns from a step's end to its send's end, spans dropped,
// tokens (live rows) of the steps served here, spans waiting to be read.
void dlia_replay_stats(void* h, uint64_t* out8) {
  Loop* l = static_cast<Loop*>(h);
  tl_reap(l);
  out8[0] = l->native_steps; out8[1] = l->handed_up; out8[2] = l->tl.busy_ns; out8[3] = l->tl.recv_wait_ns;
  out8[4] = l->tl.send_ns; out8[5] = l->tl.dropped; out8[6] = l->tl.tokens; out8[7] = l->tl.done_count;
}

// Timeline on / off (on by default: a few event records per step).
void dlia_replay_timeline(void* h, int on) { static_cast<Loop*>(h)->tl.on = on != 0; }

// Pops up to `max_spans` finished spans into `out` (kSpanDoubles doubles each: step id, micro-batch,
// rows, then microseconds on the CLOCK_MONOTONIC time base: host enqueue begin / end, stream free,
// compute begin, compute end, send end; -1 where a stamp does not apply). Returns the count.
int dlia_replay_read_spans(void* h, double* out, int max_spans) {
  Loop* l = static_cast<Loop*>(h);
  tl_reap(l);
  Timeline& t = l->tl;
  int n = 0;
  while (n < max_spans && t.done_count > 0) {
    memcpy(out + (size_t)n * kSpanDoubles, &t.done[t.done_head * kSpanDoubles], sizeof(double) * kSpanDoubles);
    t.done_head = (t.done_head + 1) % kDoneSpans; t.done_count--; ++n;
  }
  return n;
}

// Test hook support: the n-th replay step from now (n >= 1) is handed up to Python instead of being
// served here, so that a step-counting fault injector fires at the same step with or without this loop.
void dlia_replay_handoff_in(void* h, long n) { static_cast<Loop*>(h)->handoff_in = n; }

// Serves the descriptor ring until a message arrives that Python has to handle (returns 1, message
// in msg_out / *len), `idle_ms` pass without one (0), or something fails (-1, dlia_replay_error; -2:
// the descriptor ring was closed). Results of finished steps leave while it waits.
int dlia_replay_serve(void* h, void* cin_ring, void* res_ring, void* msg_out, uint32_t cap, uint32_t* len, int idle_ms) {
  Loop* l = static_cast<Loop*>(h);
  if (hipSetDevice(l->device) != hipSuccess) return -1;
  auto t0 = std::chrono::steady_clock::now();
  const auto t_enter = t0;
  int idle_spins = 0, since_check = 0;
  for (;;) {
    if (drain(l, res_ring) != 0) return -1;
    tl_reap(l);
    // back to the caller every idle_ms also under load, so that its heartbeat stays fresh while steps
    // never stop coming (queued results stay queued here and leave on the next call)
    if (idle_ms >= 0 && ++since_check >= 64) {
      since_check = 0;
      if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_enter).count() >= idle_ms)
        return 0;
    }
    const int n = l->rget(cin_ring, msg_out, cap, 0);
    if (n == -3) return -2;                                   // closed and drained
    if (n == -2) { snprintf(l->err, sizeof(l->err), "descriptor larger than the message buffer"); return -1; }
    if (n < 0) {                                              // nothing there
      if (++idle_spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(l->pending.empty() ? 100 : 20));
      else std::this_thread::yield();
      if (idle_ms >= 0 && l->pending.empty() &&
          std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >= idle_ms)
        return 0;
      continue;
    }
    idle_spins = 0;
    t0 = std::chrono::steady_clock::now();
    const DescHead* d = static_cast<const DescHead*>(msg_out);
    bool native = (size_t)n >= sizeof(DescHead) && d->magic == kDescMagic && d->kind == kKindStep &&
                  (d->flags & kFlagCompact) && (d->flags & kFlagReplay) && d->mb >= 0 && d->mb < (int)l->mbs.size() &&
                  d->ngrow >= 0 && d->ngrow <= kMaxGrow && (size_t)n >= sizeof(DescHead) + (size_t)d->ngrow * 12;
    if (native) {
      const Mb& m = l->mbs[d->mb];
      native = m.exec != nullptr && m.rows == d->rows && (d->ngrow == 0 || l->table != nullptr);
    }
    if (native && l->handoff_in > 0 && --l->handoff_in == 0) { native = false; l->handoff_in = -1; }
    if (!native) {
      *len = (uint32_t)n;
      l->handed_up++;
      return 1;
    }
    Mb& m = l->mbs[d->mb];
    StepRec* rec = tl_open(l, d->step_id, d->mb, d->rows, d->live, m.stream);
    // block-table growth: table[slot][pos] = block, checked like the Python path (a bad entry is skipped)
    const int32_t* g = reinterpret_cast<const int32_t*>(static_cast<const uint8_t*>(msg_out) + sizeof(DescHead));
    for (int i = 0; i < d->ngrow; ++i) {
      const int slot = g[i], pos = g[d->ngrow + i], blk = g[2 * d->ngrow + i];
      if (slot < 0 || slot >= l->table_rows || pos < 0 || pos >= l->max_blocks || blk < 0 || blk >= l->num_blocks) continue;
      int32_t* src = l->grow_stage + (l->grow_pos++ & 4095);
      *src = blk;
      HIPCHK(l, hipMemcpyAsync(l->table + (size_t)slot * l->max_blocks + pos, src, 4, hipMemcpyHostToDevice, m.stream));
    }
    const int s = m.next; m.next = (m.next + 1) % kSlots;
    if (hipEventQuery(m.ev[s]) == hipErrorNotReady) HIPCHK(l, hipEventSynchronize(m.ev[s]));   // slot still in flight (never in practice)
    tl_stamp(l, rec, 1, m.stream);
    HIPCHK(l, hipGraphLaunch(m.exec, m.stream));
    tl_stamp(l, rec, 2, m.stream);
    HIPCHK(l, hipMemcpyAsync(m.host[s], m.tokens, (size_t)m.rows * 4, hipMemcpyDeviceToHost, m.stream));
    HIPCHK(l, hipEventRecord(m.ev[s], m.stream));
    l->pending.push_back(Pending{d->step_id, m.rows, m.host[s], m.ev[s]});
    if (rec) rec->host1_ns = now_ns();
    l->native_steps++;
  }
}

// ---- a stage behind stage 0 ------------------------------------------------------------------------
// The hops of this stage: `in` = the receiving end of the hop from the previous stage, `out` = the
// sending end of the hop to the next stage (on the last stage: of the token feedback hop to stage 0).
int dlia_stage_set_edges(void* h, void* in_h, void* in_recv, void* in_wait, void* in_err, void* out_h, void* out_send,
                         void* out_wait, void* out_err, void* out_stream) {
  Loop* l = static_cast<Loop*>(h);
  l->out_stream = (hipStream_t)out_stream;
  l->ein.h = in_h; l->ein.recv = (edge_recv_fn)in_recv; l->ein.wait = (edge_wait_fn)in_wait; l->ein.err = (edge_err_fn)in_err;
  l->eout.h = out_h; l->eout.send = (edge_send_fn)out_send; l->eout.wait = (edge_wait_fn)out_wait; l->eout.err = (edge_err_fn)out_err;
  return 0;
}

static int hop_fail(Loop* l, const char* what, const EdgeRef& e) {
  snprintf(l->err, sizeof(l->err), "hop: %s: %s", what, e.err ? e.err() : "?");
  return -1;
}

// Registers the captured decode step of micro-batch `mb` on this stage: graph, stream, its static
// input buffer (where the receive lands) and output buffer (what is sent on). exec == null unregisters.
int dlia_stage_set_mb(void* h, int mb, void* exec, void* stream, void* in_ptr, size_t in_bytes, const void* out_ptr,
                      size_t out_bytes, int rows) {
  Loop* l = static_cast<Loop*>(h);
  if (mb < 0 || mb >= (int)l->mbs.size()) { snprintf(l->err, sizeof(l->err), "micro-batch %d out of range", mb); return -1; }
  Mb& m = l->mbs[mb];
  m.exec = (hipGraphExec_t)exec; m.stream = (hipStream_t)stream; m.rows = rows;
  m.in_ptr = in_ptr; m.in_bytes = in_bytes; m.out_ptr = out_ptr; m.out_bytes = out_bytes;
  m.in_free_ok = false;
  return 0;
}

// Before control goes back to the Python loop: the micro-batch streams wait for the sends this loop
// issued out of their output buffers, so that whatever runs on them next may overwrite those.
static int stage_flush(Loop* l) {
  for (Mb& m : l->mbs) {
    if (m.send_tok >= 0 && m.stream != nullptr) {
      if (l->eout.wait(l->eout.h, m.send_tok, m.stream) != 0) return hop_fail(l, "wait for a send", l->eout);
      m.send_tok = -1;
    }
    if (m.fb_tok >= 0 && m.stream != nullptr) {               // stage 0: the ids the micro-batch's next step reads
      if (l->ein.wait(l->ein.h, m.fb_tok, m.stream) != 0) return hop_fail(l, "wait for the fed-back ids", l->ein);
      m.fb_tok = -1;
    }
  }
  return 0;
}

// Serves the previous stage's descriptor ring. Returns like dlia_replay_serve: 1 = a message the
// Python loop has to handle (in msg_out / *len), 0 = `idle_ms` passed, -2 = ring closed, -1 = error.
// `next_ring` is null on the last stage.
int dlia_stage_serve(void* h, void* prev_ring, void* next_ring, void* msg_out, uint32_t cap, uint32_t* len, int idle_ms) {
  Loop* l = static_cast<Loop*>(h);
  if (hipSetDevice(l->device) != hipSuccess) return -1;
  if (!l->ein.h || !l->eout.h) { snprintf(l->err, sizeof(l->err), "stage loop without hops"); return -1; }
  // whatever the Python loop ran since the last call may have read the input buffers too: the first receive of
  // every micro-batch waits for its whole stream again
  for (Mb& m : l->mbs) m.in_free_ok = false;
  const auto t_enter = std::chrono::steady_clock::now();
  int idle_spins = 0, since_check = 0;
  for (;;) {
    tl_reap(l);
    if (idle_ms >= 0 && ++since_check >= 64) {
      since_check = 0;
      if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_enter).count() >= idle_ms) {
        return stage_flush(l) != 0 ? -1 : 0;
      }
    }
    const int n = l->rget(prev_ring, msg_out, cap, 0);
    if (n == -3) return -2;
    if (n == -2) { snprintf(l->err, sizeof(l->err), "descriptor larger than the message buffer"); return -1; }
    if (n < 0) {
      if (++idle_spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));
      else std::this_thread::yield();
      continue;
    }
    idle_spins = 0;
    const DescHead* d = static_cast<const DescHead*>(msg_out);
    bool native = (size_t)n >= sizeof(DescHead) && d->magic == kDescMagic && d->kind == kKindStep &&
                  (d->flags & kFlagCompact) && (d->flags & kFlagReplay) && d->mb >= 0 && d->mb < (int)l->mbs.size() &&
                  d->ngrow >= 0 && d->ngrow <= kMaxGrow && (size_t)n >= sizeof(DescHead) + (size_t)d->ngrow * 12;
    if (native) {
      const Mb& m = l->mbs[d->mb];
      native = m.exec != nullptr && m.rows == d->rows && m.in_ptr != nullptr && (d->ngrow == 0 || l->table != nullptr);
    }
    if (native && l->handoff_in > 0 && --l->handoff_in == 0) { native = false; l->handoff_in = -1; }
    if (!native) {
      if (stage_flush(l) != 0) return -1;
      *len = (uint32_t)n;
      l->handed_up++;
      return 1;
    }
    Mb& m = l->mbs[d->mb];
    StepRec* rec = tl_open(l, d->step_id, d->mb, d->rows, d->live, m.stream);
    const int32_t* g = reinterpret_cast<const int32_t*>(static_cast<const uint8_t*>(msg_out) + sizeof(DescHead));
    for (int i = 0; i < d->ngrow; ++i) {
      const int slot = g[i], pos = g[d->ngrow + i], blk = g[2 * d->ngrow + i];
      if (slot < 0 || slot >= l->table_rows || pos < 0 || pos >= l->max_blocks || blk < 0 || blk >= l->num_blocks) continue;
      int32_t* src = l->grow_stage + (l->grow_pos++ & 4095);
      *src = blk;
      HIPCHK(l, hipMemcpyAsync(l->table + (size_t)slot * l->max_blocks + pos, src, 4, hipMemcpyHostToDevice, m.stream));
    }
    // the receive overwrites the input buffer: after its previous reader — this loop's last launch of the
    // micro-batch's graph when that is known, the stream's earlier work otherwise
    const int rtok = m.in_free_ok ? l->ein.recv(l->ein.h, m.in_ptr, m.in_bytes, DLIA_NO_STREAM, m.in_free)
                                  : l->ein.recv(l->ein.h, m.in_ptr, m.in_bytes, m.stream, nullptr);
    tl_stamp(l, rec, 0, m.stream);                            // the stream is free for this step from here on
    if (rtok < 0 || l->ein.wait(l->ein.h, rtok, m.stream) != 0) return hop_fail(l, "receive", l->ein);
    if (m.send_tok >= 0) {                                    // the graph overwrites the output buffer: after its last send
      if (l->eout.wait(l->eout.h, m.send_tok, m.stream) != 0) return hop_fail(l, "wait for a send", l->eout);
      m.send_tok = -1;
    }
    tl_stamp(l, rec, 1, m.stream);                            // input there, output buffer free: compute starts
    HIPCHK(l, hipGraphLaunch(m.exec, m.stream));
    tl_stamp(l, rec, 2, m.stream);
    if (!m.in_free) HIPCHK(l, hipEventCreateWithFlags(&m.in_free, hipEventDisableTiming));
    HIPCHK(l, hipEventRecord(m.in_free, m.stream));
    m.in_free_ok = true;
    if (next_ring != nullptr && l->rput(next_ring, msg_out, (uint32_t)n, 5000) != 0) {   // descriptor first: the next stage posts its receive
      snprintf(l->err, sizeof(l->err), "next stage's descriptor ring: put failed");
      return -1;
    }
    m.send_tok = l->eout.send(l->eout.h, m.out_ptr, m.out_bytes, m.stream);
    if (m.send_tok < 0) return hop_fail(l, "send", l->eout);
    if (l->out_stream != nullptr) tl_stamp(l, rec, 3, l->out_stream);
    if (rec) rec->host1_ns = now_ns();
    l->native_steps++;
  }
}

// ---- stage 0 of a multi-stage pipeline ---------------------------------------------------------------
// Hops: dlia_stage_set_edges with `in` = the receiving end of the token feedback hop and `out` = the
// sending end of the hop to stage 1. Micro-batches: dlia_replay_set_mb (graph, stream, the graph's token
// buffer, rows) followed by dlia_head_set_out (the graph's hidden-state output).
int dlia_head_set_out(void* h, int mb, const void* out_ptr, size_t out_bytes) {
  Loop* l = static_cast<Loop*>(h);
  if (mb < 0 || mb >= (int)l->mbs.size()) { snprintf(l->err, sizeof(l->err), "micro-batch %d out of range", mb); return -1; }
  Mb& m = l->mbs[mb];
  m.out_ptr = out_ptr; m.out_bytes = out_bytes;
  m.in_ptr = const_cast<int32_t*>(m.tokens); m.in_bytes = (size_t)m.rows * 4;
  m.send_tok = -1; m.fb_tok = -1; m.last_slot = -1;          // whatever Python left pending is on the stream already
  return 0;
}

// Serves the scheduler's descriptor ring on stage 0 of several. Returns like dlia_replay_serve. Before a
// message goes up to Python every micro-batch stream has been made to wait for the ids and sends this
// loop left pending on it, so the Python loop needs no handles from here.
int dlia_head_serve(void* h, void* cin_ring, void* res_ring, void* next_ring, void* msg_out, uint32_t cap, uint32_t* len,
                    int idle_ms) {
  Loop* l = static_cast<Loop*>(h);
  if (hipSetDevice(l->device) != hipSuccess) return -1;
  if (!l->ein.h || !l->eout.h || !next_ring) { snprintf(l->err, sizeof(l->err), "head loop without hops"); return -1; }
  if (!l->res_stream) HIPCHK(l, hipStreamCreateWithFlags(&l->res_stream, hipStreamNonBlocking));
  auto t0 = std::chrono::steady_clock::now();
  const auto t_enter = t0;
  int idle_spins = 0, since_check = 0;
  for (;;) {
    if (drain(l, res_ring) != 0) return -1;
    tl_reap(l);
    if (idle_ms >= 0 && ++since_check >= 64) {
      since_check = 0;
      if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_enter).count() >= idle_ms)
        return stage_flush(l) != 0 ? -1 : 0;
    }
    const int n = l->rget(cin_ring, msg_out, cap, 0);
    if (n == -3) return -2;
    if (n == -2) { snprintf(l->err, sizeof(l->err), "descriptor larger than the message buffer"); return -1; }
    if (n < 0) {
      if (++idle_spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(l->pending.empty() ? 100 : 20));
      else std::this_thread::yield();
      if (idle_ms >= 0 && l->pending.empty() &&
          std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >= idle_ms)
        return stage_flush(l) != 0 ? -1 : 0;
      continue;
    }
    idle_spins = 0;
    t0 = std::chrono::steady_clock::now();
    const DescHead* d = static_cast<const DescHead*>(msg_out);
    bool native = (size_t)n >= sizeof(DescHead) && d->magic == kDescMagic && d->kind == kKindStep &&
                  (d->flags & kFlagCompact) && (d->flags & kFlagReplay) && d->mb >= 0 && d->mb < (int)l->mbs.size() &&
                  d->ngrow >= 0 && d->ngrow <= kMaxGrow && (size_t)n >= sizeof(DescHead) + (size_t)d->ngrow * 12;
    if (native) {
      const Mb& m = l->mbs[d->mb];
      native = m.exec != nullptr && m.rows == d->rows && m.in_ptr != nullptr && m.out_ptr != nullptr &&
               (d->ngrow == 0 || l->table != nullptr);
    }
    if (native && l->handoff_in > 0 && --l->handoff_in == 0) { native = false; l->handoff_in = -1; }
    if (!native) {
      if (stage_flush(l) != 0) return -1;
      *len = (uint32_t)n;
      l->handed_up++;
      return 1;
    }
    Mb& m = l->mbs[d->mb];
    StepRec* rec = tl_open(l, d->step_id, d->mb, d->rows, d->live, m.stream);
    const int32_t* g = reinterpret_cast<const int32_t*>(static_cast<const uint8_t*>(msg_out) + sizeof(DescHead));
    for (int i = 0; i < d->ngrow; ++i) {
      const int slot = g[i], pos = g[d->ngrow + i], blk = g[2 * d->ngrow + i];
      if (slot < 0 || slot >= l->table_rows || pos < 0 || pos >= l->max_blocks || blk < 0 || blk >= l->num_blocks) continue;
      int32_t* src = l->grow_stage + (l->grow_pos++ & 4095);
      *src = blk;
      HIPCHK(l, hipMemcpyAsync(l->table + (size_t)slot * l->max_blocks + pos, src, 4, hipMemcpyHostToDevice, m.stream));
    }
    tl_stamp(l, rec, 0, m.stream);
    if (m.fb_tok >= 0) {                                      // the ids this step embeds: fed back by the last stage
      if (l->ein.wait(l->ein.h, m.fb_tok, m.stream) != 0) return hop_fail(l, "wait for the fed-back ids", l->ein);
      m.fb_tok = -1;
    }
    if (m.send_tok >= 0) {                                    // the graph overwrites the output buffer: after its last send
      if (l->eout.wait(l->eout.h, m.send_tok, m.stream) != 0) return hop_fail(l, "wait for a send", l->eout);
      m.send_tok = -1;
    }
    tl_stamp(l, rec, 1, m.stream);
    HIPCHK(l, hipGraphLaunch(m.exec, m.stream));
    tl_stamp(l, rec, 2, m.stream);
    if (l->rput(next_ring, msg_out, (uint32_t)n, 5000) != 0) {      // descriptor first: the next stage posts its receive
      snprintf(l->err, sizeof(l->err), "next stage's descriptor ring: put failed");
      return -1;
    }
    m.send_tok = l->eout.send(l->eout.h, m.out_ptr, m.out_bytes, m.stream);
    if (m.send_tok < 0) return hop_fail(l, "send", l->eout);
    if (l->out_stream != nullptr) tl_stamp(l, rec, 3, l->out_stream);
    // This step's ids come back from the last stage into the token buffer: behind the graph that still
    // reads it (stream order) and behind the copy that took the previous ids out of it.
    const int s = m.next; m.next = (m.next + 1) % kSlots;
    if (hipEventQuery(m.ev[s]) == hipErrorNotReady) HIPCHK(l, hipEventSynchronize(m.ev[s]));
    const int rtok = l->ein.recv(l->ein.h, m.in_ptr, m.in_bytes, m.stream, m.last_slot >= 0 ? m.ev[m.last_slot] : nullptr);
    if (rtok < 0 || l->ein.wait(l->ein.h, rtok, l->res_stream) != 0) return hop_fail(l, "receive of the fed-back ids", l->ein);
    HIPCHK(l, hipMemcpyAsync(m.host[s], m.in_ptr, m.in_bytes, hipMemcpyDeviceToHost, l->res_stream));
    HIPCHK(l, hipEventRecord(m.ev[s], l->res_stream));
    l->pending.push_back(Pending{d->step_id, m.rows, m.host[s], m.ev[s]});
    m.fb_tok = rtok; m.last_slot = s;
    if (rec) rec->host1_ns = now_ns();
    l->native_steps++;
  }
}

// Blocks until every queued result has left (or `timeout_ms` passed: returns 1).
int dlia_replay_flush(void* h, void* res_ring, int timeout_ms) {
  Loop* l = static_cast<Loop*>(h);
  auto t0 = std::chrono::steady_clock::now();
  while (!l->pending.empty()) {
    if (drain(l, res_ring) != 0) return -1;
    if (l->pending.empty()) break;
    if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >= timeout_ms) return 1;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  return 0;
}

void dlia_replay_destroy(void* h) {
  Loop* l = static_cast<Loop*>(h);
  for (Mb& m : l->mbs)
    for (int i = 0; i < kSlots; ++i) {
      if (m.host[i]) (void)hipHostFree(m.host[i]);
      if (m.ev[i]) (void)hipEventDestroy(m.ev[i]);
    }
  for (Mb& m : l->mbs) if (m.in_free) (void)hipEventDestroy(m.in_free);
  if (l->grow_stage) (void)hipHostFree(l->grow_stage);
  for (StepRec& r : l->tl.open)
    for (hipEvent_t e : r.e) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : l->tl.pool) (void)hipEventDestroy(e);
  for (Anchor& a : l->tl.anchors) if (a.ev) (void)hipEventDestroy(a.ev);
  if (l->res_stream) (void)hipStreamDestroy(l->res_stream);
  delete l;
}

}  // extern "C"
