// Rotary embedding + paged KV-cache write, one launch per layer.
//
// Input is the output row of the fused QKV projection, [T, (Hq + 2 Hkv) * D] bf16. The q slice
// is rotated in place; the k slice is rotated and written, together with the v slice, into the
// layer's paged cache at (block_table[seq][pos / 32], pos % 32) in MFMA-fragment order
// (kv_layout.h). cos/sin come from a table built once per stage.
//
// The reference has no KV cache (`use_cache=False`, Worker1.py:134) and recomputes rotary
// tables for the full sequence on every worker for every token (Worker1.py:93-117).
#include "common.h"
#include "kv_layout.h"
#include "launch.h"

namespace dlia {

template <bool ROPE>
__global__ __launch_bounds__(256) void rope_kvwrite_kernel(
    bf16* __restrict__ qkv, bf16* __restrict__ kv, const float* __restrict__ cos_t,
    const float* __restrict__ sin_t, const int* __restrict__ tok_pos, const int* __restrict__ tok_seq,
    const int* __restrict__ block_table, int Hq, int Hkv, int D, int max_blocks, int num_blocks, int rope_len) {
  const int t = blockIdx.x;
  const int pos = tok_pos[t];
  const int seq = tok_seq[t];
  const int bidx = pos / kBlockTokens;
  const int kk = pos % kBlockTokens;
  int blk = -1;
  if (pos >= 0 && seq >= 0 && bidx < max_blocks) blk = block_table[(long)seq * max_blocks + bidx];
  const bool kv_ok = blk >= 0 && blk < num_blocks;   // an unallocated slot is skipped, never a wild store
  const int half = D >> 1, octs = half >> 3;
  bf16* row = qkv + (long)t * (Hq + 2 * Hkv) * D;

  const int nrot = (Hq + Hkv) * octs;
  for (int it = threadIdx.x; it < nrot; it += 256) {
    const int head = it / octs, o = it - head * octs;
    bf16* p = row + head * D + o * 8;
    const bool is_k = head >= Hq;
    if (!ROPE && !is_k) continue;
    float lo[8], hi[8];
    unpack8(ld16(p), lo);
    unpack8(ld16(p + half), hi);
    if (ROPE) {
      const int rp = pos < 0 ? 0 : (pos >= rope_len ? rope_len - 1 : pos);   // table reads stay in bounds
      const float* cp = cos_t + (long)rp * half + o * 8;
      const float* sp = sin_t + (long)rp * half + o * 8;
      const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
      const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c = j < 4 ? c0[j] : c1[j - 4], s = j < 4 ? s0[j] : s1[j - 4];
        const float a = lo[j], b = hi[j];
        lo[j] = a * c - b * s;
        hi[j] = b * c + a * s;
      }
    }
    const u32x4 vlo = pack8(lo), vhi = pack8(hi);
    if (!is_k) {
      st16(p, vlo);
      st16(p + half, vhi);
    } else if (kv_ok) {
      bf16* base = kv + kv_head_base(blk, head - Hq, Hkv, D, 0);
      st16(base + k_elem_off(kk, o * 8, D), vlo);
      st16(base + k_elem_off(kk, half + o * 8, D), vhi);
    }
  }
  if (!kv_ok) return;
  const int vocts = D >> 3;
  const int nv = Hkv * vocts;
  const unsigned short* vrow = reinterpret_cast<const unsigned short*>(row + (Hq + Hkv) * D);
  unsigned short* kvs = reinterpret_cast<unsigned short*>(kv);
  for (int it = threadIdx.x; it < nv; it += 256) {
    const int h = it / vocts, o = it - h * vocts;
    const u32x4 v = ld16(vrow + h * D + o * 8);
    unsigned short* base = kvs + kv_head_base(blk, h, Hkv, D, 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned w = v[j >> 1];
      base[v_elem_off(kk, o * 8 + j)] = (unsigned short)((j & 1) ? (w >> 16) : (w & 0xffffu));
    }
  }
}

}  // namespace dlia

extern "C" void dlia_rope_kvwrite(void* qkv, void* kv_cache, const float* cos_t, const float* sin_t,
                                  const int* tok_pos, const int* tok_seq, const int* block_table,
                                  int T, int Hq, int Hkv, int D, int max_blocks, int num_blocks,
                                  int rope_len, int use_rope, hipStream_t stream) {
  if (T <= 0) return;
  using namespace dlia;
  if (use_rope)
    hipLaunchKernelGGL((rope_kvwrite_kernel<true>), dim3(T), dim3(256), 0, stream, (bf16*)qkv, (bf16*)kv_cache,
                       cos_t, sin_t, tok_pos, tok_seq, block_table, Hq, Hkv, D, max_blocks, num_blocks, rope_len);
  else
    hipLaunchKernelGGL((rope_kvwrite_kernel<false>), dim3(T), dim3(256), 0, stream, (bf16*)qkv, (bf16*)kv_cache,
                       cos_t, sin_t, tok_pos, tok_seq, block_table, Hq, Hkv, D, max_blocks, num_blocks, rope_len);
}
