// Rotary embedding + paged KV-cache write, one launch per layer.
//
// Input is the output row of the fused QKV projection, [T, (Hq + 2 Hkv) * D] bf16. The q slice
// is rotated in place; the k slice is rotated and written, together with the v slice, into the
// layer's paged cache at (block_table[seq][pos / 32], pos % 32) in MFMA-fragment order
// (kv_layout.h). cos/sin come from a table built once per stage.
//
// KV8: the cache is OCP e4m3 with one fp32 scale per (token, kv head, K|V) (kv_layout.h). The items
// of one head are consecutive threads of an aligned lane group (D/16 lanes for K, D/8 for V), so
// the head's max |x| is a few shuffles; values are quantised from fp32, not from the bf16 rounding.
//
// The reference has no KV cache (`use_cache=False`, Worker1.py:134) and recomputes rotary
// tables for the full sequence on every worker for every token (Worker1.py:93-117).
#include "common.h"
#include "kv_layout.h"
#include "launch.h"

namespace dlia {

// Loads 8 consecutive columns [col, col+8) of token row t: from the bf16 activation, or — when the
// QKV GEMM ran split-K — as the sum of its fp32 partials (+ bias), which folds that reduction
// into this kernel.
struct QkvSrc {
  const bf16* row;            // bf16 path: this token's row
  const float* partial;       // fp32 [S][T][W] or null
  const bf16* bias;
  int splits; long T; long t; int W;
  __device__ __forceinline__ void load8(int col, float (&f)[8]) const {
    if (!partial) { unpack8(ld16(row + col), f); return; }
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = 0.f;
    const float* pr = partial + (long)t * W + col;
    const long ss = T * (long)W;                     // stride between splits
    int s = 0;
    for (; s + 4 <= splits; s += 4) {                // 8 independent 16-B loads in flight
      f32x4 a[4], c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = *reinterpret_cast<const f32x4*>(pr + (s + u) * ss);
        c[u] = *reinterpret_cast<const f32x4*>(pr + (s + u) * ss + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f[0] += a[u][0]; f[1] += a[u][1]; f[2] += a[u][2]; f[3] += a[u][3];
        f[4] += c[u][0]; f[5] += c[u][1]; f[6] += c[u][2]; f[7] += c[u][3];
      }
    }
    for (; s < splits; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(pr + s * ss), c = *reinterpret_cast<const f32x4*>(pr + s * ss + 4);
      f[0] += a[0]; f[1] += a[1]; f[2] += a[2]; f[3] += a[3];
      f[4] += c[0]; f[5] += c[1]; f[6] += c[2]; f[7] += c[3];
    }
    if (bias) {
      float d[8];
      unpack8(ld16(bias + col), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += d[j];
    }
  }
};

// 8 fp32 values / scale -> 8 e4m3 bytes
__device__ __forceinline__ u32x2 quant8(const float (&f)[8], float inv) {
  unsigned lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
  return u32x2{lo, hi};
}

template <bool ROPE, bool KV8>
__global__ __launch_bounds__(256) void rope_kvwrite_kernel(
    bf16* __restrict__ qkv, const float* __restrict__ partial, int splits, const bf16* __restrict__ pbias, int T,
    bf16* __restrict__ kv, float* __restrict__ kv_scale, const float* __restrict__ cos_t,
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
  const QkvSrc src{row, partial, pbias, splits, (long)T, (long)t, (Hq + 2 * Hkv) * D};

  const int nrot = (Hq + Hkv) * octs;
  const int tstep = 256 * gridDim.y;
  for (int it = threadIdx.x + 256 * blockIdx.y; it < nrot; it += tstep) {
    const int head = it / octs, o = it - head * octs;
    bf16* p = row + head * D + o * 8;
    const bool is_k = head >= Hq;
    if (!ROPE && !is_k && !partial) continue;     // q is already in place and needs no rotation
    float lo[8], hi[8];
    src.load8(head * D + o * 8, lo);
    src.load8(head * D + o * 8 + half, hi);
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
    if (!is_k) {
      st16(p, pack8(lo));
      st16(p + half, pack8(hi));
    } else if constexpr (KV8) {
      float am = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) am = fmaxf(am, fmaxf(fabsf(lo[j]), fabsf(hi[j])));
      for (int x = 1; x < octs; x <<= 1) am = fmaxf(am, __shfl_xor(am, x, 64));   // the head's octs lanes
      const float sc = fmaxf(am, 1e-12f) * (1.f / 448.f), inv = 1.f / sc;
      if (kv_ok) {
        unsigned char* base = reinterpret_cast<unsigned char*>(kv) + kv_head_base(blk, head - Hq, Hkv, D, 0);
        *reinterpret_cast<u32x2*>(base + k8_byte_off(kk, o * 8, D)) = quant8(lo, inv);
        *reinterpret_cast<u32x2*>(base + k8_byte_off(kk, half + o * 8, D)) = quant8(hi, inv);
        if (o == 0) kv_scale[kv_scale_base(blk, head - Hq, Hkv, 0) + kk] = sc;
      }
    } else if (kv_ok) {
      bf16* base = kv + kv_head_base(blk, head - Hq, Hkv, D, 0);
      st16(base + k_elem_off(kk, o * 8, D), pack8(lo));
      st16(base + k_elem_off(kk, half + o * 8, D), pack8(hi));
    }
  }
  if (!kv_ok) return;
  const int vocts = D >> 3;
  const int nv = Hkv * vocts;
  unsigned short* kvs = reinterpret_cast<unsigned short*>(kv);
  for (int it = threadIdx.x + 256 * blockIdx.y; it < nv; it += tstep) {
    const int h = it / vocts, o = it - h * vocts;
    float vf[8];
    src.load8((Hq + Hkv) * D + h * D + o * 8, vf);
    if constexpr (KV8) {
      float am = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(vf[j]));
      for (int x = 1; x < vocts; x <<= 1) am = fmaxf(am, __shfl_xor(am, x, 64));   // the head's vocts lanes
      const float sc = fmaxf(am, 1e-12f) * (1.f / 448.f);
      const u32x2 q = quant8(vf, 1.f / sc);
      unsigned char* base = reinterpret_cast<unsigned char*>(kv) + kv_head_base(blk, h, Hkv, D, 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) base[v8_byte_off(kk, o * 8 + j)] = (unsigned char)(q[j >> 2] >> ((j & 3) * 8));
      if (o == 0) kv_scale[kv_scale_base(blk, h, Hkv, 1) + kk] = sc;
    } else {
      const u32x4 v = pack8(vf);
      unsigned short* base = kvs + kv_head_base(blk, h, Hkv, D, 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned w = v[j >> 1];
        base[v_elem_off(kk, o * 8 + j)] = (unsigned short)((j & 1) ? (w >> 16) : (w & 0xffffu));
      }
    }
  }
}

}  // namespace dlia

// kv_scale != null: kv_cache is the e4m3 cache (bytes) and kv_scale its [blocks][Hkv][2][32] scales.
extern "C" void dlia_rope_kvwrite(void* qkv, const float* partial, int splits, const void* pbias, void* kv_cache,
                                  float* kv_scale, const float* cos_t, const float* sin_t,
                                  const int* tok_pos, const int* tok_seq, const int* block_table,
                                  int T, int Hq, int Hkv, int D, int max_blocks, int num_blocks,
                                  int rope_len, int use_rope, hipStream_t stream) {
  if (T <= 0) return;
  using namespace dlia;
  // few tokens (decode): spread each token's heads over several workgroups so the chip is not idle
  const int items = (Hq + Hkv) * (D / 16) + Hkv * (D / 8);
  int gy = 1;
  while (gy < 8 && T * gy < 512 && items > 256 * gy) gy *= 2;
#define DLIA_ROPE(R, Q)                                                                                               \
  hipLaunchKernelGGL((rope_kvwrite_kernel<R, Q>), dim3(T, gy), dim3(256), 0, stream, (bf16*)qkv, partial, splits,     \
                     (const bf16*)pbias, T, (bf16*)kv_cache, kv_scale, cos_t, sin_t, tok_pos, tok_seq, block_table,  \
                     Hq, Hkv, D, max_blocks, num_blocks, rope_len)
  if (kv_scale) { if (use_rope) DLIA_ROPE(true, true); else DLIA_ROPE(false, true); }
  else          { if (use_rope) DLIA_ROPE(true, false); else DLIA_ROPE(false, false); }
#undef DLIA_ROPE
}
