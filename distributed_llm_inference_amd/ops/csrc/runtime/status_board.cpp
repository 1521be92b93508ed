// Shared-memory status board: one cache-line-sized record per pipeline stage, written by the
// stage (heartbeat, state, progress) and read by the orchestrator's /workers and /metrics
// handlers and by the watchdog. This is the in-node replacement for the reference's liveness
// probe, which is an HTTP GET to each worker's /health with a 5 s timeout
// (orchestration.py:306-329).
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <fcntl.h>
#include <new>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace {

struct alignas(128) StageRecord {
  std::atomic<int64_t> heartbeat_ns;   // CLOCK_MONOTONIC of the last beat
  std::atomic<int64_t> steps_done;
  std::atomic<int64_t> tokens_done;
  std::atomic<int64_t> busy_ns;        // accumulated time inside step execution
  std::atomic<int32_t> pid;
  std::atomic<int32_t> state;          // 0 unregistered, 1 starting, 2 ready, 3 error, 4 stopped
  int32_t layer_start, layer_end;
  char error[64];
  std::atomic<int64_t> kv_blocks;      // KV blocks this stage's HBM budget allows (0 = not known yet)
};
static_assert(sizeof(StageRecord) == 128, "one record per 128-byte line");

struct Board {
  uint32_t magic;
  uint32_t num_stages;
  StageRecord rec[64];
};

struct Handle { Board* b; size_t bytes; char name[128]; };
constexpr uint32_t kMagic = 0x424F5244;

inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

void* dlia_board_open(const char* name, int create, uint32_t num_stages, int timeout_ms) {
  if (num_stages > 64) return nullptr;
  int fd = -1;
  if (create) {
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return nullptr;
    if (ftruncate(fd, sizeof(Board)) != 0) { close(fd); return nullptr; }
  } else {
    auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      fd = shm_open(name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Board)) break;
      if (fd >= 0) { close(fd); fd = -1; }
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (ms > timeout_ms) return nullptr;
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
  }
  void* p = mmap(nullptr, sizeof(Board), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return nullptr;
  Handle* h = new (std::nothrow) Handle();
  h->b = reinterpret_cast<Board*>(p);
  h->bytes = sizeof(Board);
  strncpy(h->name, name, sizeof(h->name) - 1);
  if (create) {
    // (the fresh segment is zero-filled by ftruncate; clearing it here again would wipe a registration that an
    // attaching stage made between the two)
    h->b->num_stages = num_stages;
    std::atomic_thread_fence(std::memory_order_release);
    h->b->magic = kMagic;
  } else {
    // The segment has its size before it has its contents: wait until the creator has published the header. Without
    // this a stage attaching in that window read 0 stages and "agreed" on the KV blocks of an empty list.
    auto t0 = std::chrono::steady_clock::now();
    while (reinterpret_cast<volatile uint32_t&>(h->b->magic) != kMagic) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (ms > timeout_ms) { munmap(p, sizeof(Board)); delete h; return nullptr; }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  return h;
}

int dlia_board_num_stages(void* hd) { return (int)static_cast<Handle*>(hd)->b->num_stages; }

void dlia_board_register(void* hd, int stage, int pid, int layer_start, int layer_end) {
  StageRecord& r = static_cast<Handle*>(hd)->b->rec[stage];
  r.layer_start = layer_start; r.layer_end = layer_end;
  r.pid.store(pid); r.heartbeat_ns.store(now_ns()); r.state.store(1);
}

void dlia_board_set_state(void* hd, int stage, int state, const char* err) {
  StageRecord& r = static_cast<Handle*>(hd)->b->rec[stage];
  if (err) { strncpy(r.error, err, sizeof(r.error) - 1); r.error[sizeof(r.error) - 1] = 0; }
  r.heartbeat_ns.store(now_ns());
  r.state.store(state);
}

void dlia_board_beat(void* hd, int stage, int64_t steps, int64_t tokens, int64_t busy_ns) {
  StageRecord& r = static_cast<Handle*>(hd)->b->rec[stage];
  r.steps_done.fetch_add(steps); r.tokens_done.fetch_add(tokens); r.busy_ns.fetch_add(busy_ns);
  r.heartbeat_ns.store(now_ns());
}

void dlia_board_set_kv(void* hd, int stage, int64_t blocks) {
  static_cast<Handle*>(hd)->b->rec[stage].kv_blocks.store(blocks);
}

// out[0]=state out[1]=pid out[2]=age_ms of the heartbeat out[3]=steps out[4]=tokens out[5]=busy_ns
// out[6]=layer_start out[7]=layer_end out[8]=kv_blocks
void dlia_board_read(void* hd, int stage, int64_t* out, char* err_out, int err_cap) {
  StageRecord& r = static_cast<Handle*>(hd)->b->rec[stage];
  out[0] = r.state.load(); out[1] = r.pid.load();
  out[2] = (now_ns() - r.heartbeat_ns.load()) / 1000000;
  out[3] = r.steps_done.load(); out[4] = r.tokens_done.load(); out[5] = r.busy_ns.load();
  out[6] = r.layer_start; out[7] = r.layer_end; out[8] = r.kv_blocks.load();
  if (err_out && err_cap > 0) { strncpy(err_out, r.error, err_cap - 1); err_out[err_cap - 1] = 0; }
}

void dlia_board_close(void* hd, int unlink_segment) {
  Handle* h = static_cast<Handle*>(hd);
  munmap(h->b, h->bytes);
  if (unlink_segment) shm_unlink(h->name);
  delete h;
}

}  // extern "C"
