// Causal GQA attention over the paged, fragment-ordered KV cache (kv_layout.h).
//
// paged_attn_kernel serves single-token decode, short chunks and split-KV (prefill_attn_kernel
// further down takes chunks of more than 128 query columns per sequence): a work item is
// (sequence b, kv head, query tile, kv split) and is owned by one 4-wave workgroup whose waves
// interleave the item's KV blocks and merge their (m, l, O) through LDS once at the end — the
// main loop has no LDS traffic and no barriers. paged_attn_lean_kernel is the same body with ONE
// wave per item, for decode steps with a thousand items or more: no merge, no uneven block counts
// between waves, 129 registers, so three waves per SIMD hide each other's load latency.
// The wave's MFMA columns are "query columns" c = (position-in-tile, head-in-GQA-group): all
// query heads that share a kv head ride in the same MFMA, so K/V are read once per group.
//
// Per 32-token block (both MFMAs are v_mfma_f32_16x16x32_bf16):
//   S^T[key, c]  = K[key, :] . Q[c, :]          A = K fragment (1 KiB coalesced load, as stored)
//                                               B = Q fragment (registers, loaded once)
//   online softmax down each column — the column lives in one lane's registers (+ the other
//   three lane quads), so max/sum need two shuffles, not a 16-lane tree
//   O^T[d, c]   += V^T[d, key] . P^T[key, c]    A = V fragment (as stored, key order permuted
//                                               to match), B = P^T = the S^T accumulator
//                                               registers packed to bf16, no data movement
//
// KV8: the cache holds OCP e4m3 bytes + one fp32 scale per (token, kv head, K|V) (kv_layout.h). A
// 16-byte lane load carries two fragments, which are widened to bf16 in registers (exact) right
// before use, so a block costs half the HBM bytes and half the load instructions, and the
// prefetch buffers half the registers. The K scale of a key multiplies its row of S^T; the V scale
// is folded into P^T (the softmax sum uses the unscaled probabilities).
//
// Split-KV (decode with few sequences / long contexts): each split writes unnormalised
// (m, l, O) per column and a second kernel merges them.
//
// Replaces HF SDPA inside LlamaDecoderLayer, which the reference runs over the full S x S
// problem on every step with no cache (Worker1.py:128-137, use_cache=False).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "kv_layout.h"
#include "launch.h"

namespace dlia {

struct AttnParams {
  bf16* out; long ldo;            // [T, Hq*D]
  const bf16* q; long ldq;        // q rows (post-RoPE) inside the fused qkv activation
  const bf16* kv;                 // this layer's cache (bf16, or e4m3 bytes when kv_scale is set)
  const float* kv_scale;          // [blocks][Hkv][2][32] or null
  const int* q_start;             // [B+1] first token row of each sequence's chunk
  const int* ctx_lens;            // [B] tokens already cached before this chunk
  const int* block_table;         // [table_rows, max_blocks]
  const int* slots;               // optional [B]: block_table row of each sequence (else row b)
  int table_rows;
  float* partial;                 // split-KV partials
  int B, Hq, Hkv, g_shift;        // query columns per position: 1 << g_shift, the GQA group size rounded up to a power of two
  int g_real;                     // the group size itself: columns whose head index within the group is >= g_real are padding
  int max_blocks, num_blocks, qtiles, splits;
  float scale_log2;               // softmax scale * log2(e)
  int xcd_map;                    // prefill kernel: 0 plain order, 1 group i on XCD i % 8, r > 1: r XCDs per group (see there)
};

constexpr float kNegInf = -__builtin_inff();

// NW = waves per work item: 4 interleave the item's KV blocks and merge at the end (few items, long
// contexts); 1 walks all of them alone with no LDS and no barrier (decode over thousands of
// (sequence, kv head) items, where a 160-token context is 5 blocks and 4 waves would take 2, 1, 1, 1).
template <int D, int NT, bool KV8, int NW, bool PF>
__device__ __forceinline__ void paged_attn_body(const AttnParams& p, long item) {
  constexpr int KS = D / 32;      // MFMA k-steps over the head dim (QK^T)
  constexpr int DT = D / 16;      // 16-row tiles of the head dim (PV)
  constexpr int NC = NT * 16;     // query columns per workgroup
  constexpr bool PREFETCH = NT == 1 && PF;   // decode: next block's K/V in flight while this one is used
  // LDS: per wave its (m, l) per lane and its O accumulators, merged by the 4 waves at the end
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, l15 = lane & 15, quad = lane >> 4;
  const int wave = NW == 1 ? 0 : threadIdx.x >> 6;
  const int split = (int)(item % p.splits); item /= p.splits;
  const int qtile = (int)(item % p.qtiles); item /= p.qtiles;
  const int kvh = (int)(item % p.Hkv);
  const int b = (int)(item / p.Hkv);

  const int g = 1 << p.g_shift;
  const int QT = NC >> p.g_shift;                 // query positions per tile
  const int t0 = p.q_start[b];
  const int q_len = p.q_start[b + 1] - t0;
  const int ctx = p.ctx_lens[b];
  int srow = p.slots ? p.slots[b] : b;
  const bool srow_ok = srow >= 0 && srow < p.table_rows;
  srow = srow_ok ? srow : 0;
  const int qp0 = qtile * QT;
  const bool tile_live = qp0 < q_len;             // workgroup-uniform

  // column bookkeeping: lane's column within tile nt is c = nt*16 + l15
  int col_qpos[NT];       // position within the chunk, or -1 when the column is padding
  long col_row[NT];       // token row of the column
  int col_head[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = nt * 16 + l15;
    const int pl = c >> p.g_shift, hl = c & (g - 1);
    const int qp = qp0 + pl;
    col_qpos[nt] = (tile_live && qp < q_len && hl < p.g_real) ? qp : -1;
    col_row[nt] = (long)(t0 + (qp < q_len ? qp : 0));
    col_head[nt] = kvh * p.g_real + hl;
  }

  // kv block range of this split; the 4 waves of the workgroup interleave its blocks.
  // A row that ran past its table (a finished sequence still riding in a fixed-shape batch) is
  // clamped to the table: reads stay in bounds, its output is discarded by the scheduler.
  const int tile_q_end = min(q_len, qp0 + QT);
  const int kv_len = tile_live ? min(ctx + tile_q_end, p.max_blocks * kBlockTokens) : 0;
  const int nb = (kv_len + kBlockTokens - 1) / kBlockTokens;
  const int per = (nb + p.splits - 1) / p.splits;
  const int kb0 = split * per, kb1 = min(nb, kb0 + per);

  // Q fragments (B operand): lane (c, quad) holds Q[c][s*32 + quad*8 .. +8]
  u32x4 qf[NT][KS];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (col_qpos[nt] >= 0)
        qf[nt][s] = ld16(p.q + col_row[nt] * p.ldq + col_head[nt] * D + s * 32 + quad * 8);
      else
        qf[nt][s] = u32x4{0, 0, 0, 0};
    }

  f32x4 o[DT][NT];
  float m[NT], l[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    m[nt] = kNegInf; l[nt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const long slab = kv_slab_elems(D);
  const int* table = p.block_table + (long)srow * p.max_blocks;
  auto block_of = [&](int kb) -> int {
    const int blk = table[kb];
    return (srow_ok && blk >= 0 && blk < p.num_blocks) ? blk : 0;          // never a wild read
  };
  // One block's operands as loaded. bf16: the fragments themselves. KV8: 16-byte loads holding two
  // fragments each, plus this lane's 8 key scales (keys 16 kt + 4 quad + r) for K and for V.
  struct Raw {
    u32x4 k[2][KV8 ? KS / 2 : KS];
    u32x4 v[KV8 ? DT / 2 : DT];
    f32x4 sk[2], sv[2];
  };
  auto load_kv = [&](int kb, Raw& r) {
    const int blk = block_of(kb);
    if constexpr (KV8) {
      const unsigned char* kbase = reinterpret_cast<const unsigned char*>(p.kv) + kv_head_base(blk, kvh, p.Hkv, D, 0) + lane * 16;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int t = 0; t < KS / 2; ++t) r.k[kt][t] = *reinterpret_cast<const u32x4*>(kbase + (kt * (KS / 2) + t) * 1024);
#pragma unroll
      for (int t = 0; t < DT / 2; ++t) r.v[t] = *reinterpret_cast<const u32x4*>(kbase + slab + t * 1024);
      const float* sb = p.kv_scale + kv_scale_base(blk, kvh, p.Hkv, 0) + quad * 4;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        r.sk[kt] = *reinterpret_cast<const f32x4*>(sb + kt * 16);
        r.sv[kt] = *reinterpret_cast<const f32x4*>(sb + kBlockTokens + kt * 16);
      }
    } else {
      const bf16* kbase = p.kv + kv_head_base(blk, kvh, p.Hkv, D, 0) + lane * 8;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < KS; ++s) r.k[kt][s] = ld16(kbase + kt * 16 * D + s * 512);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) r.v[dt] = ld16(kbase + slab + dt * 512);
    }
  };

  auto process = [&](int kb, const Raw& raw) {
    auto kfrag = [&](int kt, int s) -> u32x4 {
      if constexpr (KV8) return fp8x8_to_bf16x8(raw.k[kt][s >> 1][(s & 1) * 2], raw.k[kt][s >> 1][(s & 1) * 2 + 1]);
      else return raw.k[kt][s];
    };
    auto vfrag = [&](int dt) -> u32x4 {
      if constexpr (KV8) return fp8x8_to_bf16x8(raw.v[dt >> 1][(dt & 1) * 2], raw.v[dt >> 1][(dt & 1) * 2 + 1]);
      else return raw.v[dt];
    };
    // S^T = K Q^T : st[kt][nt][r] = score(key = kb*32 + kt*16 + quad*4 + r, column nt*16+l15)
    f32x4 st[2][NT];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      u32x4 kf[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) kf[s] = kfrag(kt, s);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) a = mfma16(kf[s], qf[nt][s], a);
        if constexpr (KV8) a *= raw.sk[kt];
        st[kt][nt] = a;
      }
    }
    u32x4 pf[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int limit = col_qpos[nt] >= 0 ? ctx + col_qpos[nt] : -1;   // last visible key position
      float mx = kNegInf;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kpos = kb * kBlockTokens + kt * 16 + quad * 4 + r;
          const float sc = kpos <= limit ? st[kt][nt][r] : kNegInf;
          st[kt][nt][r] = sc;
          mx = fmaxf(mx, sc);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m[nt], mx);
      const float m_safe = m_new == kNegInf ? 0.f : m_new;
      const float alpha = exp2f((m[nt] - m_safe) * p.scale_log2);     // m = -inf -> 0
      m[nt] = m_new;
      float pv[8], rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = exp2f((st[kt][nt][r] - m_safe) * p.scale_log2);
          rs += e;
          pv[kt * 4 + r] = KV8 ? e * raw.sv[kt][r] : e;       // V's per-key scale rides in P^T
        }
      l[nt] = l[nt] * alpha + rs;
      // The running maximum settles after the first blocks of a row; from then on alpha is exactly 1
      // and rescaling the DT accumulator tiles (a third of this loop's VALU work) is a no-op. The
      // prefill instantiations skip it unless some lane of the wave needs it (-4% on an 8192-token
      // prompt). The decode instantiation must not: the branch stops the compiler from overlapping
      // the prefetched K/V loads with the MFMAs and cost 13% of a whole Llama-3-8B decode step.
      bool rescale = true;
      if constexpr (NT > 1) rescale = __builtin_amdgcn_ballot_w64(alpha != 1.f) != 0;
      if (rescale) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][nt] *= alpha;
      }
      pf[nt] = pack8(pv);     // element j: key 16*(j>>2) + 4*quad + (j&3) — the V slab's key order
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const u32x4 vf = vfrag(dt);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[dt][nt] = mfma16(vf, pf[nt], o[dt][nt]);
    }
  };

  if constexpr (PREFETCH) {
    Raw ra, rb;
    int kb = kb0 + wave;
    if (kb < kb1) load_kv(kb, ra);
    for (; kb < kb1; kb += 2 * NW) {
      if (kb + NW < kb1) load_kv(kb + NW, rb);
      process(kb, ra);
      if (kb + NW < kb1) {
        if (kb + 2 * NW < kb1) load_kv(kb + 2 * NW, ra);
        process(kb + NW, rb);
      }
    }
  } else {
    for (int kb = kb0 + wave; kb < kb1; kb += NW) {
      Raw r;
      load_kv(kb, r);
      process(kb, r);
    }
  }

  if constexpr (NW == 1) {
    // ---- one wave owns the item: normalise and store from registers ----------------------------
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float lt = l[nt];
      lt += __shfl_xor(lt, 16, 64);
      lt += __shfl_xor(lt, 32, 64);
      if (col_qpos[nt] < 0) continue;
      const float inv = lt > 0.f ? 1.f / lt : 0.f;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const f32x4 acc = o[dt][nt];
        if (p.splits == 1) {
          u32x2 v = {pack_bf16(acc[0] * inv, acc[1] * inv), pack_bf16(acc[2] * inv, acc[3] * inv)};
          *reinterpret_cast<u32x2*>(p.out + col_row[nt] * p.ldo + col_head[nt] * D + dt * 16 + quad * 4) = v;
        } else {
          const long rec = ((((long)b * p.Hkv + kvh) * p.qtiles + qtile) * p.splits + split) * NC + nt * 16 + l15;
          float* r = p.partial + rec * (D + 4);
          if (quad == 0 && dt == 0) { r[0] = m[nt]; r[1] = lt; }
          *reinterpret_cast<f32x4*>(r + 4 + dt * 16 + quad * 4) = acc;
        }
      }
    }
    return;
  }
  // ---- merge the 4 waves through LDS -----------------------------------------------------
  // layout: ml[w][nt][lane] = (m, l partial of this lane); oacc[w][dt][nt][lane] f32x4
  float* ml = reinterpret_cast<float*>(smem);                         // 4 * NT * 64 * 2 floats
  f32x4* oacc = reinterpret_cast<f32x4*>(smem + 4 * NT * 64 * 2 * 4);  // 4 * DT * NT * 64 f32x4
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float lt = l[nt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    ml[((wave * NT + nt) * 64 + lane) * 2] = m[nt];
    ml[((wave * NT + nt) * 64 + lane) * 2 + 1] = lt;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[((wave * DT + dt) * NT + nt) * 64 + lane] = o[dt][nt];
  }
  __syncthreads();
  // wave w merges the head-dim tiles dt = w, w+4, ...
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float mw[4], lw[4], mt = kNegInf;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      mw[w] = ml[((w * NT + nt) * 64 + lane) * 2];
      lw[w] = ml[((w * NT + nt) * 64 + lane) * 2 + 1];
      mt = fmaxf(mt, mw[w]);
    }
    const float m_safe = mt == kNegInf ? 0.f : mt;
    float wgt[4], lt = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      wgt[w] = exp2f((mw[w] - m_safe) * p.scale_log2);      // -inf -> 0
      lt += lw[w] * wgt[w];
    }
    if (col_qpos[nt] < 0) continue;
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    for (int dt = wave; dt < DT; dt += 4) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w) acc += oacc[((w * DT + dt) * NT + nt) * 64 + lane] * wgt[w];
      if (p.splits == 1) {
        u32x2 v = {pack_bf16(acc[0] * inv, acc[1] * inv), pack_bf16(acc[2] * inv, acc[3] * inv)};
        *reinterpret_cast<u32x2*>(p.out + col_row[nt] * p.ldo + col_head[nt] * D + dt * 16 + quad * 4) = v;
      } else {
        // record = [m, l, pad, pad, O[D]] fp32 per (item, column), unnormalised
        const long rec = ((((long)b * p.Hkv + kvh) * p.qtiles + qtile) * p.splits + split) * NC + nt * 16 + l15;
        float* r = p.partial + rec * (D + 4);
        if (quad == 0 && dt == wave) { r[0] = mt; r[1] = lt; }
        *reinterpret_cast<f32x4*>(r + 4 + dt * 16 + quad * 4) = acc;
      }
    }
  }
}

template <int D, int NT, bool KV8>
__global__ __launch_bounds__(256) void paged_attn_kernel(AttnParams p) { paged_attn_body<D, NT, KV8, 4, true>(p, blockIdx.x); }

// One wave per item, one block in flight per wave, registers capped so that 3 waves share a SIMD:
// latency is hidden by the other waves instead of by a second set of K/V registers.
template <int D, bool KV8>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4))) void paged_attn_lean_kernel(AttnParams p) {
  paged_attn_body<D, 1, KV8, 1, false>(p, blockIdx.x);
}

// The same one-wave-per-item body in a fixed number of 4-wave workgroups (one wave per SIMD) that walk the item list
// with a stride: an experiment switch (DLIA_ATTN_PERSIST=<workgroups>, off by default). A CU that runs a workgroup of
// the four-wave GEMM (368 of a SIMD lane's 512 registers) has room for exactly one of these, so with one workgroup per
// CU attention waves of one micro-batch can sit beside GEMM waves of the other instead of taking turns with them.
template <int D, bool KV8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void paged_attn_lean_persist_kernel(AttnParams p, long nitems) {
  const long first = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (long item = first; item < nitems; item += (long)gridDim.x * 4) paged_attn_body<D, 1, KV8, 1, false>(p, item);
}

// Merge split-KV partials. One workgroup per (b, kv head, query tile); thread -> (column, 4 dims).
template <int D, int NT>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnParams p) {
  constexpr int NC = NT * 16;
  long item = blockIdx.x;
  const int qtile = (int)(item % p.qtiles); item /= p.qtiles;
  const int kvh = (int)(item % p.Hkv);
  const int b = (int)(item / p.Hkv);
  const int g = 1 << p.g_shift;
  const int QT = NC >> p.g_shift;
  const int t0 = p.q_start[b];
  const int q_len = p.q_start[b + 1] - t0;
  const long rec0 = (((long)b * p.Hkv + kvh) * p.qtiles + qtile) * p.splits * NC;
  for (int w = threadIdx.x; w < NC * (D / 4); w += 256) {
    const int c = w / (D / 4), d4 = (w - c * (D / 4)) * 4;
    const int qp = qtile * QT + (c >> p.g_shift);
    if (qp >= q_len || (c & (g - 1)) >= p.g_real) continue;
    float mt = kNegInf;
    for (int s = 0; s < p.splits; ++s) mt = fmaxf(mt, p.partial[(rec0 + (long)s * NC + c) * (D + 4)]);
    float lt = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (mt != kNegInf) {
      for (int s = 0; s < p.splits; ++s) {
        const float* r = p.partial + (rec0 + (long)s * NC + c) * (D + 4);
        const float ms = r[0];
        if (ms == kNegInf) continue;
        const float wgt = exp2f((ms - mt) * p.scale_log2);
        lt += r[1] * wgt;
        const f32x4 ov = *reinterpret_cast<const f32x4*>(r + 4 + d4);
        acc += ov * wgt;
      }
    }
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    const int head = kvh * p.g_real + (c & (g - 1));
    u32x2 v = {pack_bf16(acc[0] * inv, acc[1] * inv), pack_bf16(acc[2] * inv, acc[3] * inv)};
    *reinterpret_cast<u32x2*>(p.out + (long)(t0 + qp) * p.ldo + head * D + d4) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Prefill-shaped variant, used for chunks of more than 128 query columns per sequence.
//
// The kernel above gives every wave ALL query columns of the workgroup and a quarter of the KV
// blocks, which is right for decode (few columns, the KV stream is the cost) and wrong for long
// chunks: at 64 columns a wave holds 128 accumulator + 64 Q + 128 K/V registers, the compiler
// parks the excess in AGPRs and a fifth of the kernel is v_accvgpr moves; the 131 KB merge buffer
// leaves one wave per SIMD; counters on an 8192-token prompt show 12 VALU instructions per MFMA,
// VALU active 59 % of resident wave time and the matrix cores 17 %.
//
// Here the 4 waves split the QUERY columns instead (32 each, 128 per workgroup) and share every
// KV block through LDS. A (block, kv head) is K slab + V slab back to back in memory and already
// in fragment order, so staging it is a linear copy (16 B per thread and piece) and a fragment
// is one conflict-free ds_read_b128 at fragment*1 KiB + lane*16 (an e4m3 cache is widened to bf16
// while it is staged, once per workgroup). Block k+1 is loaded global -> registers while block k
// is computed, then written to the other buffer: one barrier per block.
// No cross-wave merge, 33 KB of LDS, K/V are fetched once per 128 columns.
// Softmax work per score is cut to what the hardware needs: raw v_exp_f32 (results below the
// normal range flush to zero, which a probability may), one FMA instead of subtract + multiply,
// the causal mask only in blocks that can reach the wave's diagonal, the accumulator rescale only
// when some running maximum moved.
// Exchange with lane^16 / lane^32 on the VALU (v_permlane16_swap: odd 16-lane rows of the first operand
// <-> even rows of the second; v_permlane32_swap: upper half <-> lower half). With both operands = x
// the two results hold, in every lane, its own value and its partner's, so a max or a sum over the
// four lanes of a column is two swaps — no LDS round trip in the middle of the softmax chain, unlike
// ds_bpermute.
__device__ __forceinline__ float quad_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int D, bool KV8>
__global__ __launch_bounds__(256, 2) void prefill_attn_kernel(AttnParams p) {
  constexpr int KS = D / 32, DT = D / 16, NT = 2, NC = 128;
  constexpr int BLOCK_BYTES = 2 * 2 * kBlockTokens * D;               // K slab + V slab of one head as bf16 (in LDS)
  constexpr int MEM_BYTES = KV8 ? BLOCK_BYTES / 2 : BLOCK_BYTES;      // ... and as stored in the cache
  constexpr int PIECES = MEM_BYTES / (256 * 16);
  static_assert(PIECES >= 1 && PIECES * 256 * 16 == MEM_BYTES, "a block is whole 16-byte pieces per thread");
  constexpr int BUF = BLOCK_BYTES + 256;                              // + the block's 64 fp32 scales (KV8)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, quad = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroup -> (sequence, kv head, query tile), longest KV prefix first. Workgroups go to the 8 XCDs
  // round-robin by blockIdx and every XCD has its own L2. All query tiles of one (sequence, kv head)
  // stream the same K/V, so with xcd_map (the number of such groups is a multiple of 8) group i runs
  // entirely on XCD i % 8: an XCD's resident workgroups then walk ONE head's blocks at about the same
  // time and hit in its L2, instead of every L2 seeing every head.
  // xcd_map > 1 (fewer than 8 groups, their number divides 8): every group owns xcd_map XCDs and its tiles
  // are dealt over them, heaviest first; the grid is padded to whole rounds, surplus workgroups leave.
  int qtile, kvh, b;
  if (p.xcd_map == 1) {
    const int w = blockIdx.x >> 3;
    const int grp = (blockIdx.x & 7) + 8 * (w / p.qtiles);
    qtile = p.qtiles - 1 - (w % p.qtiles);
    kvh = grp % p.Hkv;
    b = grp / p.Hkv;
  } else if (p.xcd_map > 1) {
    const int x = blockIdx.x & 7, w = blockIdx.x >> 3;
    const int grp = x / p.xcd_map;
    qtile = p.qtiles - 1 - (w * p.xcd_map + x % p.xcd_map);
    if (qtile < 0) return;                        // whole workgroup, before any barrier
    kvh = grp % p.Hkv;
    b = grp / p.Hkv;
  } else {
    long item = blockIdx.x;
    qtile = p.qtiles - 1 - (int)(item % p.qtiles); item /= p.qtiles;
    kvh = (int)(item % p.Hkv);
    b = (int)(item / p.Hkv);
  }

  const int g = 1 << p.g_shift;
  const int QT = NC >> p.g_shift;                 // query positions per workgroup
  const int t0 = p.q_start[b];
  const int q_len = p.q_start[b + 1] - t0;
  const int ctx = p.ctx_lens[b];
  int srow = p.slots ? p.slots[b] : b;
  const bool srow_ok = srow >= 0 && srow < p.table_rows;
  srow = srow_ok ? srow : 0;
  const int qp0 = qtile * QT;
  const bool tile_live = qp0 < q_len;             // workgroup-uniform

  // Column c = wave*32 + nt*16 + l15 is (query position qp0 + (c >> g_shift), head c & (g-1) of the group).
  // Only the position stays in registers across the loop (the causal limit); token row and head are
  // recomputed where they are needed, before and after it.
  int col_qpos[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int qp = qp0 + ((wave * 32 + nt * 16 + l15) >> p.g_shift);
    col_qpos[nt] = (tile_live && qp < q_len && ((wave * 32 + nt * 16 + l15) & (g - 1)) < p.g_real) ? qp : -1;
  }
  // wave-uniform: first and last query position of this wave's columns
  const int wq_first = qp0 + ((wave * 32) >> p.g_shift);
  const int wq_last = min(q_len - 1, qp0 + ((wave * 32 + 31) >> p.g_shift));
  const bool wave_live = tile_live && wq_first < q_len;
  const int lim_last = ctx + wq_last;             // largest

  const int tile_q_end = min(q_len, qp0 + QT);
  const int kv_len = tile_live ? min(ctx + tile_q_end, p.max_blocks * kBlockTokens) : 0;
  const int nb = (kv_len + kBlockTokens - 1) / kBlockTokens;      // workgroup-uniform

  u32x4 qf[NT][KS];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (col_qpos[nt] >= 0)
        qf[nt][s] = ld16(p.q + (long)(t0 + col_qpos[nt]) * p.ldq + (kvh * p.g_real + ((wave * 32 + nt * 16 + l15) & (g - 1))) * D + s * 32 + quad * 8);
      else
        qf[nt][s] = u32x4{0, 0, 0, 0};
    }
  f32x4 o[DT][NT];
  float m[NT], l[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    m[nt] = kNegInf; l[nt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int* table = p.block_table + (long)srow * p.max_blocks;
  u32x4 stage[PIECES];
  f32x4 sstage = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int blk) {
    blk = (srow_ok && blk >= 0 && blk < p.num_blocks) ? blk : 0;            // never a wild read
    const char* src = reinterpret_cast<const char*>(p.kv) + kv_head_base(blk, kvh, p.Hkv, D, 0) * (KV8 ? 1 : 2);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) stage[i] = *reinterpret_cast<const u32x4*>(src + (i * 256 + tid) * 16);
    if constexpr (KV8) {
      if (tid < 16) sstage = *reinterpret_cast<const f32x4*>(p.kv_scale + kv_scale_base(blk, kvh, p.Hkv, 0) + tid * 4);
    }
  };
  // KV8: a 16-byte piece of the cache is lane `lane`'s part of bf16 fragments 2R and 2R+1 (R = the
  // piece's 1 KiB run, K runs first, then V). It is widened HERE, once per workgroup, so the four
  // waves read plain bf16 fragments and the compute path is the bf16 one plus the two scale multiplies.
  auto lstore = [&](int boff) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      if constexpr (KV8) {
        const int run = i * 4 + (tid >> 6);
        *reinterpret_cast<u32x4*>(smem + boff + (2 * run) * 1024 + lane * 16) = fp8x8_to_bf16x8(stage[i][0], stage[i][1]);
        *reinterpret_cast<u32x4*>(smem + boff + (2 * run + 1) * 1024 + lane * 16) = fp8x8_to_bf16x8(stage[i][2], stage[i][3]);
      } else {
        *reinterpret_cast<u32x4*>(smem + boff + (i * 256 + tid) * 16) = stage[i];
      }
    }
    if constexpr (KV8) {
      if (tid < 16) *reinterpret_cast<f32x4*>(smem + boff + BLOCK_BYTES + tid * 16) = sstage;
    }
  };

  auto process = [&](int kb, int boff, auto mask_c) {
    constexpr bool MASK = decltype(mask_c)::value;
    f32x4 sk[2], sv[2];
    if constexpr (KV8) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        sk[kt] = *reinterpret_cast<const f32x4*>(smem + boff + BLOCK_BYTES + (kt * 16 + quad * 4) * 4);
        sv[kt] = *reinterpret_cast<const f32x4*>(smem + boff + BLOCK_BYTES + (kBlockTokens + kt * 16 + quad * 4) * 4);
      }
    }
    f32x4 st[2][NT];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      u32x4 kf[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) kf[s] = *reinterpret_cast<const u32x4*>(smem + boff + (kt * KS + s) * 1024 + lane * 16);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) a = mfma16(kf[s], qf[nt][s], a);
        if constexpr (KV8) a *= sk[kt];
        st[kt][nt] = a;
      }
    }
    u32x4 pf[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float mx = kNegInf;
      if constexpr (MASK) {
        const int limit = col_qpos[nt] >= 0 ? ctx + col_qpos[nt] : -1;   // last visible key position
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kpos = kb * kBlockTokens + kt * 16 + quad * 4 + r;
            const float sc = kpos <= limit ? st[kt][nt][r] : kNegInf;
            st[kt][nt][r] = sc;
            mx = fmaxf(mx, sc);
          }
      } else {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][nt][r]);
      }
      mx = quad_max(mx);
      const float m_new = fmaxf(m[nt], mx);
      const float m_safe = m_new == kNegInf ? 0.f : m_new;
      const float ms = m_safe * p.scale_log2;
      const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m[nt], p.scale_log2, -ms));     // m = -inf -> 0
      m[nt] = m_new;
      float pv[8], rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][nt][r], p.scale_log2, -ms));   // masked: -inf -> 0
          rs += e;
          pv[kt * 4 + r] = KV8 ? e * sv[kt][r] : e;       // V's per-key scale rides in P^T
        }
      l[nt] = l[nt] * alpha + rs;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][nt] *= alpha;
      }
      pf[nt] = pack8(pv);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const u32x4 vf = *reinterpret_cast<const u32x4*>(smem + boff + BLOCK_BYTES / 2 + dt * 1024 + lane * 16);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[dt][nt] = mfma16(vf, pf[nt], o[dt][nt]);
    }
  };

  // The block table is read 64 entries at a time, one per lane; a block's entry is then a v_readlane:
  // no memory access (and no wait on a memory counter) inside the loop except once per 64 blocks.
  int tbl = 0;
  auto tbl_fill = [&](int kb) { tbl = (kb + lane < p.max_blocks) ? table[kb + lane] : 0; };
  auto tbl_at = [&](int kb) { return __builtin_amdgcn_readlane(tbl, kb & 63); };
  tbl_fill(0);
  if (nb > 0) { gload(tbl_at(0)); lstore(0); }
  // Everything loaded so far (the Q fragments, block 0) has landed before the loop starts: vmcnt is
  // one in-order counter, so a wait for a prologue load placed inside the loop would also wait for
  // the prefetch of the next block that was issued just before it.
  __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
  __syncthreads();
  // Two loops instead of one loop choosing between two inlined copies of process(): blocks that end at
  // or below the smallest causal limit in the workgroup (wave 0's first column) need no mask in any
  // wave; the last two or three blocks run the masked form, which is also right for a block that
  // happens to need none. With both forms in one loop the accumulators lived in different registers
  // in each and the compiler reconciled them with a dozen v_mov_b64 per block; VALU and MFMA time add
  // up in this kernel (66 % + 32 % of SIMD time), so those copies were not free.
  const int nb_full = tile_live ? min(nb, (ctx + qp0 + 1) / kBlockTokens) : 0;      // workgroup-uniform
  auto run = [&](int kb_begin, int kb_end, auto mask_c) {
    constexpr bool MASK = decltype(mask_c)::value;
    for (int kb = kb_begin; kb < kb_end; ++kb) {
      const int boff = (kb & 1) * BUF;
      if (kb + 1 < nb) {
        if (((kb + 1) & 63) == 0) tbl_fill(kb + 1);
        gload(tbl_at(kb + 1));
      }
      // masked loop: a block whose first key is beyond the wave's last column is skipped altogether
      if (wave_live && (!MASK || kb * kBlockTokens <= lim_last)) process(kb, boff, mask_c);
      if (kb + 1 < nb) lstore(BUF - boff);
      __syncthreads();
    }
  };
  run(0, nb_full, std::false_type{});
  run(nb_full, nb, std::true_type{});

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float lt = quad_sum(l[nt]);
    if (col_qpos[nt] < 0) continue;
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    int le = lane;
    asm volatile("" : "+v"(le));                   // recompute from the lane id instead of keeping the prologue's values live
    const int ce = wave * 32 + nt * 16 + (le & 15);
    bf16* orow = p.out + (long)(t0 + col_qpos[nt]) * p.ldo + (kvh * p.g_real + (ce & (g - 1))) * D + (le >> 4) * 4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x4 acc = o[dt][nt];
      u32x2 v = {pack_bf16(acc[0] * inv, acc[1] * inv), pack_bf16(acc[2] * inv, acc[3] * inv)};
      *reinterpret_cast<u32x2*>(orow + dt * 16) = v;
    }
  }
}

template <int D>
static void launch_prefill_attn(const AttnParams& p, hipStream_t stream) {
  const long nitems = p.xcd_map > 1 ? 8L * ((p.qtiles + p.xcd_map - 1) / p.xcd_map) : (long)p.B * p.Hkv * p.qtiles;
  constexpr size_t smem = 2 * (4 * kBlockTokens * D + 256);      // two bf16 blocks + their scales
  if (p.kv_scale) hipLaunchKernelGGL((prefill_attn_kernel<D, true>), dim3((unsigned)nitems), dim3(256), smem, stream, p);
  else            hipLaunchKernelGGL((prefill_attn_kernel<D, false>), dim3((unsigned)nitems), dim3(256), smem, stream, p);
}

// Work items from which one wave per item beats four (same box, Llama-3-8B heads, 150 cached
// tokens: 1024 sequences 118 vs 130 us, 256: 30.6 vs 37.0, 128: 17.3 vs 21.3, 64: 13.1 vs 11.9; 64
// sequences x 2000 tokens: 122 vs 94; e4m3 cache at 1024 sequences: 71 vs 101 us).
// DLIA_ATTN_WAVE_ITEMS overrides the threshold for A/B runs (0: always four waves per item).
static inline long wave_per_item_threshold() {
  static const long v = [] { const char* e = getenv("DLIA_ATTN_WAVE_ITEMS"); return e ? atol(e) : 1024L; }();
  return v;
}

template <int D, int NT>
static void launch_attn(const AttnParams& p, hipStream_t stream) {
  const long nitems = (long)p.B * p.Hkv * p.qtiles * p.splits;
  constexpr size_t smem = 4 * NT * 64 * 2 * 4 + 4 * (D / 16) * NT * 64 * 16;
  const long thr = wave_per_item_threshold();
  const char* pe = getenv("DLIA_ATTN_PERSIST");               // read per launch (a captured graph keeps what it was captured with)
  const long persist = pe ? atol(pe) : 0L;
  if (NT == 1 && thr > 0 && nitems >= thr && persist > 0 && nitems >= persist * 4 && !p.kv_scale) {
    hipLaunchKernelGGL((paged_attn_lean_persist_kernel<D, false>), dim3((unsigned)persist), dim3(256), 0, stream, p, nitems);
  } else if (NT == 1 && thr > 0 && nitems >= thr) {
    if (p.kv_scale) hipLaunchKernelGGL((paged_attn_lean_kernel<D, true>), dim3((unsigned)nitems), dim3(64), 0, stream, p);
    else            hipLaunchKernelGGL((paged_attn_lean_kernel<D, false>), dim3((unsigned)nitems), dim3(64), 0, stream, p);
  } else if (p.kv_scale) hipLaunchKernelGGL((paged_attn_kernel<D, NT, true>), dim3((unsigned)nitems), dim3(256), smem, stream, p);
  else                   hipLaunchKernelGGL((paged_attn_kernel<D, NT, false>), dim3((unsigned)nitems), dim3(256), smem, stream, p);
  if (p.splits > 1)
    hipLaunchKernelGGL((attn_combine_kernel<D, NT>), dim3((unsigned)((long)p.B * p.Hkv * p.qtiles)), dim3(256), 0,
                       stream, p);
}

}  // namespace dlia

using namespace dlia;

// Number of 16-column query tiles per workgroup of paged_attn_kernel: 1 for decode, 2 for chunks.
static inline int attn_nt(int max_q_len, int g) { return (max_q_len * g > 16) ? 2 : 1; }

// Chunks of more than 128 query columns per sequence without split-KV go to the prefill-shaped
// kernel (DLIA_ATTN_PREFILL=0 keeps them on paged_attn_kernel, DLIA_ATTN_XCD_MAP=0 turns the XCD-aware
// workgroup order off, both for A/B runs). One 8192-token Llama-3-8B prompt, whole model: 126.0 ms
// against 163.5 ms on paged_attn_kernel at 32 columns per workgroup (158.4 ms at 64, a variant since
// removed); 16 x 512 tokens: 106.3 against 111.8 ms.
static inline bool use_prefill_kernel(int max_q_len, int g, int splits) {
  static const bool on = [] { const char* e = getenv("DLIA_ATTN_PREFILL"); return !(e && atoi(e) == 0); }();
  return on && splits <= 1 && g <= 128 && max_q_len * g > 128;
}

// GQA group size rounded up to a power of two: the number of query columns a position takes in a tile
// (group sizes 3, 5, 6, 7 ride with padding columns; K/V are still read once per group).
static inline int padded_group(int Hq, int Hkv) {
  int g = 1;
  while (g < Hq / Hkv) g <<= 1;
  return g;
}

extern "C" long dlia_paged_attention_ws_floats(int B, int max_q_len, int Hq, int Hkv, int D, int splits) {
  if (splits <= 1) return 0;
  const int g = padded_group(Hq, Hkv);
  const int nt = attn_nt(max_q_len, g);
  const int qt = (nt * 16) / g;
  const long qtiles = (max_q_len + qt - 1) / qt;
  return (long)B * Hkv * qtiles * splits * nt * 16 * (D + 4);
}

extern "C" void dlia_paged_attention(void* out, const void* qkv, const void* kv_cache, const float* kv_scale,
                                     const int* q_start, const int* ctx_lens, const int* block_table,
                                     const int* slots, float* partial, int B, int max_q_len, int Hq,
                                     int Hkv, int D, int table_rows, int max_blocks, int num_blocks,
                                     long ldq, long ldo, float scale, int splits, hipStream_t stream) {
  if (B <= 0 || max_q_len <= 0) return;
  AttnParams p;
  p.out = (bf16*)out; p.ldo = ldo; p.q = (const bf16*)qkv; p.ldq = ldq; p.kv = (const bf16*)kv_cache; p.kv_scale = kv_scale;
  p.q_start = q_start; p.ctx_lens = ctx_lens; p.block_table = block_table; p.slots = slots; p.table_rows = table_rows; p.partial = partial;
  p.B = B; p.Hq = Hq; p.Hkv = Hkv;
  p.g_real = Hq / Hkv;
  const int g = padded_group(Hq, Hkv);
  int gs = 0;
  while ((1 << gs) < g) ++gs;
  p.g_shift = gs;
  p.max_blocks = max_blocks; p.num_blocks = num_blocks;
  p.splits = (splits > 1 && partial) ? splits : 1;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.xcd_map = 0;
  if (use_prefill_kernel(max_q_len, g, p.splits) && (D == 128 || D == 64)) {
    p.qtiles = (max_q_len + 128 / g - 1) / (128 / g);
    static const bool xcd_on = [] { const char* e = getenv("DLIA_ATTN_XCD_MAP"); return !(e && atoi(e) == 0); }();
    const long groups = (long)B * Hkv;
    p.xcd_map = !xcd_on ? 0 : (groups % 8 == 0 ? 1 : ((groups < 8 && 8 % groups == 0) ? (int)(8 / groups) : 0));
    if (D == 128) launch_prefill_attn<128>(p, stream); else launch_prefill_attn<64>(p, stream);
    return;
  }
  const int nt = attn_nt(max_q_len, g);
  const int qt = (nt * 16) / g;
  p.qtiles = (max_q_len + qt - 1) / qt;
  if (D == 128) { if (nt == 1) launch_attn<128, 1>(p, stream); else launch_attn<128, 2>(p, stream); }
  else if (D == 64) { if (nt == 1) launch_attn<64, 1>(p, stream); else launch_attn<64, 2>(p, stream); }
}
