// Single-producer / single-consumer message ring in POSIX shared memory.
//
// The control plane of the pipeline on one node: stage i hands the step descriptor of each
// micro-batch to stage i+1 through one of these (the tensors themselves travel GPU-to-GPU over
// xGMI). A descriptor is a few hundred bytes and must arrive in ~a microsecond, ahead of the
// activation it describes, so this is a lock-free ring with acquire/release indices, not a
// socket. It replaces the reference's only control path — a full HTTP request per hop
// (orchestration.py:114-137).
//
// Layout: [Header | slots x (u32 length + payload[slot_bytes])]. head = next slot to write
// (producer-owned), tail = next slot to read (consumer-owned), both monotonically increasing.
#include <atomic>
#include <cerrno>
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

struct alignas(64) Header {
  uint32_t magic;
  uint32_t slots;
  uint32_t slot_bytes;
  uint32_t closed;
  alignas(64) std::atomic<uint64_t> head;
  alignas(64) std::atomic<uint64_t> tail;
};

struct Ring {
  Header* h;
  uint8_t* data;
  size_t map_bytes;
  int owner;
  char name[128];
};

constexpr uint32_t kMagic = 0x52494E47;  // "RING"

inline uint8_t* slot_ptr(Ring* r, uint64_t idx) {
  return r->data + (idx % r->h->slots) * (size_t)(r->h->slot_bytes + 8);
}

}  // namespace

extern "C" {

// create != 0: create (or reset) the segment; else attach to an existing one (retrying briefly).
void* dlia_ring_open(const char* name, int create, uint32_t slots, uint32_t slot_bytes, int timeout_ms) {
  int fd = -1;
  if (create) {
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return nullptr;
    size_t bytes = sizeof(Header) + (size_t)slots * (slot_bytes + 8);
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return nullptr; }
  } else {
    auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      fd = shm_open(name, O_RDWR, 0600);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Header)) break;
        close(fd); fd = -1;
      }
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (ms > timeout_ms) return nullptr;
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
  }
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return nullptr; }
  size_t bytes = (size_t)st.st_size;
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return nullptr;
  Ring* r = new (std::nothrow) Ring();
  if (!r) { munmap(p, bytes); return nullptr; }
  r->h = reinterpret_cast<Header*>(p);
  r->data = reinterpret_cast<uint8_t*>(p) + sizeof(Header);
  r->map_bytes = bytes;
  r->owner = create;
  strncpy(r->name, name, sizeof(r->name) - 1);
  if (create) {
    r->h->slots = slots; r->h->slot_bytes = slot_bytes; r->h->closed = 0;
    r->h->head.store(0, std::memory_order_relaxed);
    r->h->tail.store(0, std::memory_order_relaxed);
    std::atomic_thread_fence(std::memory_order_release);
    r->h->magic = kMagic;
  } else {
    auto t0 = std::chrono::steady_clock::now();
    while (reinterpret_cast<volatile uint32_t&>(r->h->magic) != kMagic) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (ms > timeout_ms) { munmap(p, bytes); delete r; return nullptr; }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  return r;
}

uint32_t dlia_ring_slot_bytes(void* ring) { return static_cast<Ring*>(ring)->h->slot_bytes; }

// 0 ok, 1 timed out (ring full), 2 message too large, 3 closed
int dlia_ring_put(void* ring, const void* msg, uint32_t len, int timeout_ms) {
  Ring* r = static_cast<Ring*>(ring);
  Header* h = r->h;
  if (len > h->slot_bytes) return 2;
  const uint64_t head = h->head.load(std::memory_order_relaxed);
  auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (head - h->tail.load(std::memory_order_acquire) >= h->slots) {
    if (h->closed) return 3;
    if (++spins > 2000) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (timeout_ms >= 0 && ms > timeout_ms) return 1;
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  uint8_t* s = slot_ptr(r, head);
  memcpy(s, &len, 4);
  memcpy(s + 8, msg, len);
  h->head.store(head + 1, std::memory_order_release);
  return 0;
}

// >= 0: message length copied into out; -1 timed out; -2 out buffer too small; -3 closed and drained
int dlia_ring_get(void* ring, void* out, uint32_t cap, int timeout_ms) {
  Ring* r = static_cast<Ring*>(ring);
  Header* h = r->h;
  const uint64_t tail = h->tail.load(std::memory_order_relaxed);
  auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (h->head.load(std::memory_order_acquire) == tail) {
    if (h->closed) return -3;
    if (++spins > 2000) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (timeout_ms >= 0 && ms > timeout_ms) return -1;
      // back off progressively: a busy pipeline is served in the spin phase above
      std::this_thread::sleep_for(std::chrono::microseconds(ms > 50 ? 200 : 20));
    }
  }
  const uint8_t* s = slot_ptr(r, tail);
  uint32_t len;
  memcpy(&len, s, 4);
  if (len > cap) return -2;
  memcpy(out, s + 8, len);
  h->tail.store(tail + 1, std::memory_order_release);
  return (int)len;
}

int dlia_ring_pending(void* ring) {
  Header* h = static_cast<Ring*>(ring)->h;
  return (int)(h->head.load(std::memory_order_acquire) - h->tail.load(std::memory_order_acquire));
}

void dlia_ring_mark_closed(void* ring) { static_cast<Ring*>(ring)->h->closed = 1; }

void dlia_ring_close(void* ring, int unlink_segment) {
  Ring* r = static_cast<Ring*>(ring);
  munmap(r->h, r->map_bytes);
  if (unlink_segment) shm_unlink(r->name);
  delete r;
}

}  // extern "C"
