// KV-cache block allocator: a LIFO free list over block ids [0, num_blocks) with per-owner
// accounting, so a finished (or failed) sequence returns exactly what it took. Called by the
// scheduler thread only; the mutex makes the metrics readers safe.
//
// The reference has no KV cache and therefore nothing to allocate (Worker1.py:134).
#include <cstdint>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

namespace {
struct Allocator {
  std::mutex mu;
  std::vector<int32_t> free_list;
  std::unordered_map<int64_t, std::vector<int32_t>> owned;
  int32_t num_blocks;
  int64_t peak_used = 0;
};
}  // namespace

extern "C" {

void* dlia_alloc_create(int32_t num_blocks) {
  Allocator* a = new (std::nothrow) Allocator();
  if (!a) return nullptr;
  a->num_blocks = num_blocks;
  a->free_list.reserve(num_blocks);
  for (int32_t i = num_blocks - 1; i >= 0; --i) a->free_list.push_back(i);   // pop() yields 0,1,2,...
  return a;
}

void dlia_alloc_destroy(void* h) { delete static_cast<Allocator*>(h); }

// Appends n block ids for `owner` into out. All-or-nothing: returns 0 on success, -1 when fewer
// than n blocks are free (nothing is taken).
int dlia_alloc_take(void* h, int64_t owner, int32_t n, int32_t* out) {
  Allocator* a = static_cast<Allocator*>(h);
  std::lock_guard<std::mutex> g(a->mu);
  if (n < 0 || (size_t)n > a->free_list.size()) return -1;
  auto& mine = a->owned[owner];
  for (int32_t i = 0; i < n; ++i) {
    const int32_t b = a->free_list.back();
    a->free_list.pop_back();
    mine.push_back(b);
    out[i] = b;
  }
  const int64_t used = a->num_blocks - (int64_t)a->free_list.size();
  if (used > a->peak_used) a->peak_used = used;
  return 0;
}

// Returns every block of `owner`; result = number of blocks released.
int dlia_alloc_release(void* h, int64_t owner) {
  Allocator* a = static_cast<Allocator*>(h);
  std::lock_guard<std::mutex> g(a->mu);
  auto it = a->owned.find(owner);
  if (it == a->owned.end()) return 0;
  const int n = (int)it->second.size();
  for (auto rit = it->second.rbegin(); rit != it->second.rend(); ++rit) a->free_list.push_back(*rit);
  a->owned.erase(it);
  return n;
}

int32_t dlia_alloc_free_count(void* h) {
  Allocator* a = static_cast<Allocator*>(h);
  std::lock_guard<std::mutex> g(a->mu);
  return (int32_t)a->free_list.size();
}

int32_t dlia_alloc_owned_count(void* h, int64_t owner) {
  Allocator* a = static_cast<Allocator*>(h);
  std::lock_guard<std::mutex> g(a->mu);
  auto it = a->owned.find(owner);
  return it == a->owned.end() ? 0 : (int32_t)it->second.size();
}

int64_t dlia_alloc_peak_used(void* h) {
  Allocator* a = static_cast<Allocator*>(h);
  std::lock_guard<std::mutex> g(a->mu);
  return a->peak_used;
}

}  // extern "C"
