// Token sampling on device: temperature -> top-k -> top-p -> categorical draw, one workgroup
// (16 waves) per row of fp32 logits.
//
// Semantics follow the reference sampler (orchestration.py:143-169) exactly:
//   * logits / temperature;
//   * top-k keeps every logit >= the k-th largest (ties at the k-th value survive);
//   * top-p sorts the survivors descending, takes softmax + inclusive cumsum and removes entry i
//     iff cum[i-1] > top_p (the mask is shifted right by one, so the first token that crosses
//     the threshold is kept and the top-1 is always kept);
//   * softmax over what is left, then one categorical sample.
// Differences by design: temperature <= 0 means greedy argmax (the reference divides by zero),
// and the draw uses a counter-based Philox4x32-10 keyed by (seed, step) per row, so a request is
// reproducible regardless of batching (the reference uses the global unseeded torch RNG).
//
// Method for 1 <= k <= 1024 (the serving path: k = 50): a pivot taken from a sorted 1024-element
// sample puts a few hundred values above it; one pass over the row collects everything >= pivot
// into LDS (retry with a moved pivot, then an exact 4 x 8-bit radix select, if the count misses
// [k, kCap]); bitonic sort of the candidates; cut at the k-th value; block scan for the cumulative
// mass. Since top-k runs first, top-p only ever touches the <= kCap survivors.
// Method for k <= 0 ("no top-k", orchestration.py:150) and k > 1024 on rows longer than 1024: nothing
// bounds the survivors, so nothing is sorted. The sorted order (value descending, index ascending
// among equals) is walked implicitly by radix selects over the float keys, weighted by softmax mass:
// one select finds the entry in which the cumulative mass crosses top_p (everything above it, and
// the ties up to it, is the nucleus), a second one the entry in which it crosses u x kept mass (the
// draw), and a scan in index order resolves the tie rank. Exact for any k, top_p and tie pattern.
// With chunk maxima (the lm_head GEMM's epilogue leaves, per row, the maximum of every 32-column chunk):
// the k-th largest chunk maximum is a lower bound of the k-th largest logit — each of the k chunks
// at or above it holds at least one value at or above it — so every top-k candidate lies in a chunk
// whose maximum reaches that bound. The kernel selects that bound exactly (radix select over the <=
// 8192 maxima in LDS), reads only those chunks (k x 32 values instead of the whole row: 6 KB of a
// 513 KB row at k = 50, vocabulary 128256) and hands the values at or above the bound to the same
// sort / cut / draw as before: same candidates or a superset, hence the same token. Greedy rows read
// one chunk. Tie patterns that overflow the lists fall back to the whole-row methods.
// Rows must be 16-byte aligned (row stride multiple of 4).
#include "common.h"
#include "launch.h"

namespace dlia {

constexpr int kSampThreads = 1024;
constexpr int kCap = 1024;            // candidates the sorted path holds in LDS; larger / absent k takes the full-row path
constexpr int kChunk = 32;            // columns per chunk maximum (gemm_pingpong.hip: one wave's columns of a column half)
constexpr int kMaxChunks = 8192;      // chunk maxima a row may have (vocabulary <= 262144: Qwen2's 151,936 / 152,064 fit)
constexpr int kHotCap = 1024;         // chunks that may reach the bound before the whole-row method takes over

__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned u = __builtin_bit_cast(unsigned, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // monotone: larger float -> larger key
}

__device__ __forceinline__ void philox_round(unsigned (&c)[4], const unsigned (&k)[2]) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
  const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k[0];
  const unsigned n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k[1];
  const unsigned n3 = (unsigned)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned step) {
  unsigned c[4] = {step, 0u, 0u, 0u};
  unsigned k[2] = {(unsigned)seed, (unsigned)(seed >> 32)};
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k);
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
  }
  return (float)(c[0] >> 8) * (1.0f / 16777216.0f);       // [0, 1)
}

struct SampleParams {
  int* out; const float* logits; long ldl;
  const float* temperature; const int* top_k; const float* top_p;
  const unsigned long long* seeds; const int* steps;
  int vocab;
  float* dbg_probs; int* dbg_ids; int dbg_cap;
  const float* chunk_max; long ldc; int nchunks;      // optional: maxima of the row's 32-column chunks
  // optional tail of a captured decode step, done by the thread that writes the token: the token also goes into the
  // next step's input ids, and the row's draw counter and context length advance by one
  int* ids_out; int* steps_adv; int* ctx_adv;
};

__device__ __forceinline__ void emit_token(const SampleParams& p, int r, int tok) {
  p.out[r] = tok;
  if (p.ids_out) p.ids_out[r] = tok;
  if (p.steps_adv) p.steps_adv[r] += 1;             // (every read of the counter lies before a barrier this thread has passed)
  if (p.ctx_adv) p.ctx_adv[r] += 1;
}


// ---- exact full-row path (k <= 0 or k > kCap) --------------------------------------------------
// Softmax masses are carried as 40-bit fixed point (mass 1.0 = 2^40, the row maximum) in 64-bit
// integers: integer sums do not depend on the order the LDS atomics land in, so the same row and
// seed always give the same token, and equal logits have exactly equal masses, so a tie group's mass
// is its count times one member's.
typedef unsigned long long Q;
__device__ __forceinline__ Q mass_q(float v, float zmax, float inv_t) {
  return (Q)(__expf((v - zmax) * inv_t) * 1099511627776.f);
}

struct FullShared {
  Q hm[256];              // softmax mass per radix bin
  unsigned hc[256];       // entries per radix bin
  unsigned prefix;        // key bits fixed so far
  Q t_rem;                // target mass still to walk inside the current prefix
  Q s_above;              // mass of the entries sorted before the selected key
  unsigned c_above;       // their number
  Q m_at;                 // mass and number of the entries AT the selected key (after the last pass)
  unsigned c_at;
  Q redq[16];
  float red[16];
  int redi[16];
  int found;
};

__device__ __forceinline__ Q full_block_sum(Q v, FullShared& sh, int lane, int wave) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) sh.redq[wave] = v;
  __syncthreads();
  Q t = 0;
  for (int w = 0; w < 16; ++w) t += sh.redq[w];
  return t;
}

// Walks the entries with key >= max(kfloor, ceil_key) in sorted order (key descending) until the
// cumulative mass exceeds `target`: leaves in `sh` the key of the entry that contains the target
// (prefix), the mass and count of everything sorted before that KEY (s_above, c_above) and the mass /
// count at the key itself (m_at, c_at). Of the tie group at `part_key` only `part_keep` members
// belong to the set (drop = the others' count; their mass leaves the bin they fall in). A target at
// or beyond the total mass selects the smallest key present.
__device__ void full_weighted_select(const float* row, int V, float zmax, float inv_t, unsigned kfloor, unsigned part_key,
                                     unsigned part_drop, Q part_q, unsigned ceil_key, Q target, FullShared& sh,
                                     int tid, int lane, int wave) {
  if (tid == 0) { sh.prefix = 0u; sh.t_rem = target; sh.s_above = 0; sh.c_above = 0u; }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) { sh.hm[tid] = 0; sh.hc[tid] = 0u; }
    __syncthreads();
    const unsigned prefix = sh.prefix;
    const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < V; i += kSampThreads) {
      const float v = row[i];
      const unsigned key = f2key(v);
      if (key < kfloor || key < ceil_key || (key & himask) != (prefix & himask)) continue;
      atomicAdd(&sh.hm[(key >> shift) & 255u], mass_q(v, zmax, inv_t));
      atomicAdd(&sh.hc[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0 && part_drop > 0u && (part_key & himask) == (prefix & himask)) {
      sh.hm[(part_key >> shift) & 255u] -= (Q)part_drop * part_q;
      sh.hc[(part_key >> shift) & 255u] -= part_drop;
    }
    __syncthreads();
    if (wave == 0) {
      // lane owns bins 4 lane .. 4 lane + 3; suffix sums from the top bin down
      Q m[4]; unsigned c[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) { m[b] = sh.hm[4 * lane + b]; c[b] = sh.hc[4 * lane + b]; }
      const Q mine = m[0] + m[1] + m[2] + m[3];
      const unsigned cmine = c[0] + c[1] + c[2] + c[3];
      Q suf = mine; unsigned csuf = cmine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const Q t = __shfl_down(suf, o, 64); const unsigned ct = __shfl_down(csuf, o, 64);
        if (lane + o < 64) { suf += t; csuf += ct; }
      }
      const Q t_rem = sh.t_rem;
      const Q above = suf - mine;
      const unsigned cabove = csuf - cmine;
      bool hit = cmine > 0u && above <= t_rem && t_rem < suf;
      unsigned long long hits = __ballot(hit);
      if (hits == 0ull) {                         // target at or beyond the total: the smallest key present
        const unsigned long long nonempty = __ballot(cmine > 0u);
        hit = nonempty != 0ull && lane == (int)__builtin_ctzll(nonempty);
        hits = __ballot(hit);
      }
      const int first = hits ? 63 - (int)__builtin_clzll(hits) : -1;       // the top-most hit
      if (lane == first) {
        Q acc = above; unsigned cacc = cabove; int bin = -1;
#pragma unroll
        for (int b = 3; b >= 0; --b) {
          if (bin < 0 && c[b] > 0u) {
            const bool last_nonempty = (b == 0) || (c[0] + (b > 1 ? c[1] : 0u) + (b > 2 ? c[2] : 0u) == 0u);
            if (t_rem < acc + m[b] || last_nonempty) bin = b;
            else { acc += m[b]; cacc += c[b]; }
          }
        }
        if (bin < 0) bin = 0;
        sh.prefix = prefix | ((unsigned)(4 * lane + bin) << shift);
        sh.s_above += acc;
        sh.c_above += cacc;
        sh.t_rem = t_rem > acc ? t_rem - acc : 0;
        sh.m_at = m[bin]; sh.c_at = c[bin];
      }
    }
    __syncthreads();
  }
}

// The whole draw for one row; returns (through p.out[r]) the sampled token. `k` <= 0: no top-k.
__device__ void sample_full_row(const SampleParams& p, int r, const float* row, int V, float temp, int k, float top_p,
                                FullShared& sh, unsigned* hist, unsigned& s_prefix, unsigned& s_krem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float inv_t = 1.f / temp;
  // ---- top-k floor: exact key of the k-th largest value (ties at it survive) -----------------
  unsigned kfloor = 0u;
  if (k > 0 && k < V) {
    if (tid == 0) { s_prefix = 0u; s_krem = (unsigned)k; }
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      const unsigned prefix = s_prefix;
      const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int i = tid; i < V; i += kSampThreads) {
        const unsigned key = f2key(row[i]);
        if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (wave == 0) {
        const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        const unsigned mine = h0 + h1 + h2 + h3;
        unsigned suf = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned t = __shfl_down(suf, o, 64);
          if (lane + o < 64) suf += t;
        }
        const unsigned krem = s_krem;
        const unsigned above = suf - mine;
        if (above < krem && krem <= suf) {
          unsigned cnt = above; int bin;
          if (cnt + h3 >= krem) bin = 3;
          else { cnt += h3; if (cnt + h2 >= krem) bin = 2;
            else { cnt += h2; if (cnt + h1 >= krem) bin = 1; else { cnt += h1; bin = 0; } } }
          s_prefix = prefix | ((unsigned)(4 * lane + bin) << shift);
          s_krem = krem - cnt;
        }
      }
      __syncthreads();
    }
    kfloor = s_prefix;
  }
  // ---- maximum and total mass of the surviving entries ----------------------------------------
  float zmax = -__builtin_inff();
  for (int i = tid; i < V; i += kSampThreads) zmax = fmaxf(zmax, row[i]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o, 64));
  __syncthreads();
  if (lane == 0) sh.red[wave] = zmax;
  __syncthreads();
  for (int w = 0; w < 16; ++w) zmax = fmaxf(zmax, sh.red[w]);
  Q zl = 0; Q nl = 0;
  for (int i = tid; i < V; i += kSampThreads) {
    const float v = row[i];
    if (f2key(v) >= kfloor) { zl += mass_q(v, zmax, inv_t); ++nl; }
  }
  const Q Z = full_block_sum(zl, sh, lane, wave);
  const int nsurv = (int)full_block_sum(nl, sh, lane, wave);
  // ---- top-p: the entry in which the cumulative mass crosses top_p * Z ends the nucleus -----------
  // (an entry is kept iff the mass sorted before it is <= top_p * Z: orchestration.py:159-164 removes
  // the entries whose cumulative probability exceeds top_p, shifted by one so the crossing entry stays)
  unsigned bkey = kfloor; Q kept_mass = Z, part_q = 0; unsigned part_key = 0xFFFFFFFFu, part_drop = 0u; int nkeep = nsurv;
  bool cut = false;
  if (top_p < 1.f) {
    const Q P = (Q)((double)top_p * (double)Z);
    full_weighted_select(row, V, zmax, inv_t, kfloor, 0xFFFFFFFFu, 0u, 0, 0u, P, sh, tid, lane, wave);
    const Q q_b = sh.c_at ? sh.m_at / (Q)sh.c_at : 0;                       // one boundary entry's mass (ties are exact)
    // a tie with rank t (index order) is kept iff s_above + t q_b <= P
    Q tstar = q_b > 0 ? sh.t_rem / q_b : (Q)sh.c_at - 1;
    if (tstar > (Q)sh.c_at - 1) tstar = (Q)sh.c_at - 1;
    const unsigned kept_ties = (unsigned)tstar + 1u;
    bkey = sh.prefix; cut = true;
    kept_mass = sh.s_above + (Q)kept_ties * q_b;
    nkeep = (int)sh.c_above + (int)kept_ties;
    if (kept_ties < sh.c_at) { part_key = bkey; part_drop = sh.c_at - kept_ties; part_q = q_b; }
    __syncthreads();
  }
  // ---- the draw: the entry in which the kept cumulative mass crosses u * kept_mass ----------------
  const float u = philox_uniform(p.seeds ? p.seeds[r] : 0ull, p.steps ? (unsigned)p.steps[r] : 0u);
  Q target = (Q)((double)u * (double)kept_mass);
  if (kept_mass > 0 && target >= kept_mass) target = kept_mass - 1;
  full_weighted_select(row, V, zmax, inv_t, kfloor, part_key, part_drop, part_q, cut ? bkey : 0u, target, sh, tid, lane, wave);
  const unsigned dkey = sh.prefix;
  const Q q_d = sh.c_at ? sh.m_at / (Q)sh.c_at : 0;
  Q tr = q_d > 0 ? sh.t_rem / q_d : 0;
  if (sh.c_at && tr > (Q)sh.c_at - 1) tr = (Q)sh.c_at - 1;
  const int trank = (int)tr;                   // rank, in index order, among the members at the drawn key
  __syncthreads();
  // ---- the trank-th entry, in index order, among those with key == dkey ---------------------------
  if (tid == 0) sh.found = 0;
  __syncthreads();
  int seen = 0;
  for (int base = 0; base < V; base += kSampThreads) {
    const int i = base + tid;
    const bool eq = i < V && f2key(row[i]) == dkey;
    const unsigned long long bal = __ballot(eq);
    if (lane == 0) sh.redi[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) before += sh.redi[w]; total += sh.redi[w]; }
    const int my = seen + before + __popcll(bal & ((1ull << lane) - 1ull));
    if (eq && my == trank) { emit_token(p, r, i); sh.found = 1; }
    seen += total;
    __syncthreads();
    if (sh.found || seen > trank) break;
  }
  if (p.dbg_probs && tid == 0) {
    // diagnostics of this path: ids[0] = -2 (marker), ids[1] = entries kept, probs[0] = kept mass / survivor mass
    for (int i = 0; i < p.dbg_cap; ++i) { p.dbg_probs[(long)r * p.dbg_cap + i] = 0.f; p.dbg_ids[(long)r * p.dbg_cap + i] = -1; }
    p.dbg_ids[(long)r * p.dbg_cap] = -2;
    if (p.dbg_cap > 1) p.dbg_ids[(long)r * p.dbg_cap + 1] = nkeep;
    p.dbg_probs[(long)r * p.dbg_cap] = (float)((double)kept_mass / (double)Z);
  }
}

// Key of the k-th largest of vals[0, n) (LDS; 1 <= k <= n): four 8-bit radix passes, exact.
__device__ unsigned lds_kth_key(const float* vals, int n, int k, unsigned* hist, unsigned& s_prefix, unsigned& s_krem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { s_prefix = 0u; s_krem = (unsigned)k; }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < n; i += kSampThreads) {
      const unsigned key = f2key(vals[i]);
      if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (wave == 0) {
      const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned suf = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += t;
      }
      const unsigned krem = s_krem;
      const unsigned above = suf - mine;
      if (above < krem && krem <= suf) {
        unsigned cnt = above; int bin;
        if (cnt + h3 >= krem) bin = 3;
        else { cnt += h3; if (cnt + h2 >= krem) bin = 2;
          else { cnt += h2; if (cnt + h1 >= krem) bin = 1; else { cnt += h1; bin = 0; } } }
        s_prefix = prefix | ((unsigned)(4 * lane + bin) << shift);
        s_krem = krem - cnt;
      }
    }
    __syncthreads();
  }
  return s_prefix;
}

__device__ __forceinline__ float key2f(unsigned key) {
  const unsigned u = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
  return __builtin_bit_cast(float, u);
}

__global__ __launch_bounds__(kSampThreads) void sample_kernel(SampleParams p) {
  __shared__ unsigned hist[256];
  __shared__ FullShared fsh;
  __shared__ unsigned s_prefix, s_krem, s_ncand;
  __shared__ float cval[kCap];
  __shared__ int cidx[kCap];
  __shared__ float wsum[16];
  __shared__ float s_red[16];
  __shared__ int s_redi[16];
  __shared__ float cmax[kMaxChunks];
  __shared__ unsigned short hot[kHotCap];
  __shared__ unsigned s_nhot;

  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = p.logits + (long)r * p.ldl;
  const int V = p.vocab;
  const float temp = p.temperature ? p.temperature[r] : 1.f;
  const int NC = p.chunk_max != nullptr ? p.nchunks : 0;      // > 0: chunk maxima cover the row
  if (NC > 0) {
    const float* cm = p.chunk_max + (long)r * p.ldc;
    for (int c = tid; c < NC; c += kSampThreads) cmax[c] = cm[c];
    __syncthreads();
  }

  if (temp <= 0.f && NC > 0) {
    // greedy from the chunk maxima: the row maximum is the largest of them, its lowest index lies in
    // the lowest chunk that reaches it
    float bv = -__builtin_inff();
    for (int c = tid; c < NC; c += kSampThreads) bv = fmaxf(bv, cmax[c]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bv = fmaxf(bv, __shfl_xor(bv, o, 64));
    if (lane == 0) s_red[wave] = bv;
    __syncthreads();
    for (int w = 0; w < 16; ++w) bv = fmaxf(bv, s_red[w]);
    int bc = 0x7fffffff;
    for (int c = tid; c < NC; c += kSampThreads) if (cmax[c] == bv) bc = min(bc, c);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bc = min(bc, __shfl_xor(bc, o, 64));
    if (lane == 0) s_redi[wave] = bc;
    __syncthreads();
    for (int w = 0; w < 16; ++w) bc = min(bc, s_redi[w]);
    if (wave == 0) {
      const int col = bc * kChunk + lane;
      const bool hit = bc != 0x7fffffff && lane < kChunk && col < V && row[col] == bv;
      const unsigned long long b = __ballot(hit);
      if (lane == 0) {
        const int bi = b ? bc * kChunk + (int)__builtin_ctzll(b) : 0;
        emit_token(p, r, bi);
        if (p.dbg_probs) {
          for (int i = 0; i < p.dbg_cap; ++i) { p.dbg_probs[(long)r * p.dbg_cap + i] = i == 0 ? 1.f : 0.f;
                                                p.dbg_ids[(long)r * p.dbg_cap + i] = i == 0 ? bi : -1; }
        }
      }
    }
    return;
  }
  if (temp <= 0.f) {
    // greedy: max value, lowest index among ties
    float bv = -__builtin_inff(); int bi = 0x7fffffff;
    const int nv4 = V >> 2;
    const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
    for (int i = tid; i < nv4; i += kSampThreads) {
      const f32x4 v4 = row4[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = v4[j]; const int ii = i * 4 + j;
        if (v > bv || (v == bv && ii < bi)) { bv = v; bi = ii; }
      }
    }
    for (int i = (nv4 << 2) + tid; i < V; i += kSampThreads) {
      const float v = row[i];
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_red[wave] = bv; s_redi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 16; ++w)
        if (s_red[w] > bv || (s_red[w] == bv && s_redi[w] < bi)) { bv = s_red[w]; bi = s_redi[w]; }
      emit_token(p, r, bi == 0x7fffffff ? 0 : bi);
      if (p.dbg_probs) {
        for (int i = 0; i < p.dbg_cap; ++i) { p.dbg_probs[(long)r * p.dbg_cap + i] = i == 0 ? 1.f : 0.f;
                                              p.dbg_ids[(long)r * p.dbg_cap + i] = i == 0 ? bi : -1; }
      }
    }
    return;
  }

  int k = p.top_k ? p.top_k[r] : 0;
  const float top_p = p.top_p ? p.top_p[r] : 1.f;
  if (V > kCap && (k <= 0 || k > kCap)) {          // no top-k, or more survivors than LDS holds: exact full-row walk
    sample_full_row(p, r, row, V, temp, k, top_p, fsh, hist, s_prefix, s_krem);
    return;
  }
  if (k <= 0 || k > V) k = V;                      // (V <= kCap: every entry is a candidate)
  if (tid == 0) s_ncand = 0u;

  // ---- candidate gathering ----------------------------------------------------------------
  // Fast path: a pivot from a 4096-element strided sample, chosen so that a few hundred values lie
  // above it, then ONE pass that collects everything >= pivot into LDS. If at least k and at most
  // kCap were collected, the exact top-k (ties included) is among them. Only candidates touch the
  // LDS counter, so there is no atomic contention. A pivot that turns out too high or too low is
  // moved and the pass repeated; after 4 misses an exact radix select takes over.
  bool have = false;
  if (V <= kCap) {
    if (tid < V) { cval[tid] = row[tid]; cidx[tid] = tid; }
    if (tid == 0) s_ncand = (unsigned)V;
    __syncthreads();
    have = true;
  } else if (NC > 0 && k <= NC) {
    // ---- guided by the chunk maxima: only chunks that reach the k-th largest maximum can hold a candidate ----
    const float bound = key2f(lds_kth_key(cmax, NC, k, hist, s_prefix, s_krem));
    if (tid == 0) s_nhot = 0u;
    __syncthreads();
    for (int c = tid; c < NC; c += kSampThreads)
      if (cmax[c] >= bound) {
        const unsigned slot = atomicAdd(&s_nhot, 1u);
        if (slot < (unsigned)kHotCap) hot[slot] = (unsigned short)c;
      }
    __syncthreads();
    const unsigned nhot = s_nhot;
    if (nhot <= (unsigned)kHotCap) {
      for (unsigned h = wave * 2 + (lane >> 5); h < nhot; h += 32) {      // two chunks per wave and turn
        const int col = (int)hot[h] * kChunk + (lane & 31);
        if (col < V) {
          const float v = row[col];
          if (v >= bound) {
            const unsigned slot = atomicAdd(&s_ncand, 1u);
            if (slot < (unsigned)kCap) { cval[slot] = v; cidx[slot] = col; }
          }
        }
      }
      __syncthreads();
      const unsigned n = s_ncand;
      have = n >= (unsigned)k && n <= (unsigned)kCap;
    }
    __syncthreads();
  }
  if (!have && V > kCap) {
    // Sample: every thread reads one aligned float4 (one cache line per thread: 64 KiB of traffic
    // for 4096 samples) and keeps its maximum. For the small ranks used below, the r-th largest of
    // those 1024 maxima is the r-th largest of all 4096 samples unless two of the top r share a
    // thread, which only lowers the pivot a little (more candidates, never fewer).
    const int stride = V / kSampThreads;
    {
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(row + ((tid * stride) & ~3));
      cval[tid] = fmaxf(fmaxf(v4[0], v4[1]), fmaxf(v4[2], v4[3]));
    }
    cidx[tid] = tid;
    __syncthreads();
    for (int size = 2; size <= kCap; size <<= 1)            // sort the sample, descending
      for (int st = size >> 1; st > 0; st >>= 1) {
        const int partner = tid ^ st;
        if (partner > tid) {
          const float x = cval[tid], y = cval[partner];
          const bool up = (tid & size) == 0;
          if (up ? x < y : x > y) { cval[tid] = y; cval[partner] = x; }
        }
        __syncthreads();
      }
    // Aim for ~max(4k, 384) values above the pivot: with rank ~12 of 4096 samples the count has a
    // relative spread of ~28 %, so both "fewer than k" and "more than kCap" are < 1e-5 per row
    // at k = 50 (a retry costs a whole extra pass over the row, and the slowest row sets the time).
    const int target = min(max(4 * k, 384), max(640, (k + kCap) / 2));
    const int nsamp = 4 * kSampThreads;
    int rank = min(kSampThreads - 1, max(1, (int)(((long)target * nsamp) / V) - 1));
    if (tid < 16) s_red[tid] = 0.f;
    __syncthreads();
    // candidate pivots are read before the buffer is reused for the candidates themselves
    float piv_hi = cval[max(0, rank / 4)], piv = cval[rank], piv_lo = cval[min(kSampThreads - 1, rank * 4)];
    float piv_lo2 = cval[min(kSampThreads - 1, rank * 16)];
    __syncthreads();
    for (int attempt = 0; attempt < 4 && !have; ++attempt) {
      if (tid == 0) s_ncand = 0u;
      __syncthreads();
      const int nv4 = V >> 2;
      const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
      for (int i = tid; i < nv4; i += kSampThreads) {
        const f32x4 v = row4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v[j] >= piv) {
            const unsigned slot = atomicAdd(&s_ncand, 1u);
            if (slot < (unsigned)kCap) { cval[slot] = v[j]; cidx[slot] = i * 4 + j; }
          }
      }
      for (int i = (nv4 << 2) + tid; i < V; i += kSampThreads) {
        const float v = row[i];
        if (v >= piv) {
          const unsigned slot = atomicAdd(&s_ncand, 1u);
          if (slot < (unsigned)kCap) { cval[slot] = v; cidx[slot] = i; }
        }
      }
      __syncthreads();
      const unsigned n = s_ncand;
      if (n >= (unsigned)k && n <= (unsigned)kCap) have = true;
      else if (n > (unsigned)kCap) { piv_lo = piv; piv = piv_hi; piv_hi = __builtin_inff(); }   // too many: raise
      else { piv_hi = piv; piv = piv_lo; piv_lo = piv_lo2; piv_lo2 = -__builtin_inff(); }        // too few: lower
      __syncthreads();
    }
  }
  if (!have) {
  // ---- exact radix select of the k-th largest key (fallback) ----------------------------
  if (tid == 0) { s_prefix = 0u; s_krem = (unsigned)k; s_ncand = 0u; }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < V; i += kSampThreads) {
      const unsigned key = f2key(row[i]);
      if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (wave == 0) {
      // lane owns bins 4*lane .. 4*lane+3; suffix-scan lane sums from the top
      const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned suf = mine;                       // inclusive suffix sum over lanes >= lane
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += t;
      }
      const unsigned krem = s_krem;
      const unsigned above = suf - mine;         // count in bins above this lane's bins
      const bool hit = above < krem && krem <= suf;
      if (hit) {
        unsigned cnt = above; int bin;
        if (cnt + h3 >= krem) bin = 3;
        else { cnt += h3; if (cnt + h2 >= krem) bin = 2;
          else { cnt += h2; if (cnt + h1 >= krem) bin = 1; else { cnt += h1; bin = 0; } } }
        s_prefix = prefix | ((unsigned)(4 * lane + bin) << shift);
        s_krem = krem - cnt;
      }
    }
    __syncthreads();
  }
  const unsigned kth = s_prefix;
  for (int i = tid; i < V; i += kSampThreads) {
    const float v = row[i];
    if (f2key(v) >= kth) {
      const unsigned slot = atomicAdd(&s_ncand, 1u);
      if (slot < (unsigned)kCap) { cval[slot] = v; cidx[slot] = i; }
    }
  }
  __syncthreads();
  if (s_ncand > (unsigned)kCap) {                  // ties at the k-th value: more survivors than LDS holds
    __syncthreads();
    sample_full_row(p, r, row, V, temp, k, top_p, fsh, hist, s_prefix, s_krem);
    return;
  }
  }
  int ncand = min((int)s_ncand, kCap);
  __syncthreads();
  if (tid >= ncand) { cval[tid] = -__builtin_inff(); cidx[tid] = 0x7fffffff; }
  __syncthreads();

  // ---- bitonic sort, descending by value, ascending index among equals -----------------
  // over the smallest power of two that holds the candidates (the guided gather leaves ~k of them:
  // a 64-element sort is 21 exchange steps, the full 1024-element one 55)
  int sort_n = 64;
  while (sort_n < ncand) sort_n <<= 1;
  for (int size = 2; size <= sort_n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = tid ^ stride;
      if (partner > tid && partner < sort_n) {
        const float a = cval[tid], b = cval[partner];
        const int ai = cidx[tid], bi = cidx[partner];
        const bool a_first = a > b || (a == b && ai < bi);     // a should precede b in final order
        const bool up = (tid & size) == 0;                     // this pair sorts "descending"
        if (up ? !a_first : a_first) { cval[tid] = b; cval[partner] = a; cidx[tid] = bi; cidx[partner] = ai; }
      }
      __syncthreads();
    }
  }

  // ---- top-k: keep every candidate >= the k-th largest (ties at the k-th value survive) ----
  {
    const float kth_val = cval[min(k, ncand) - 1];
    const bool in_k = tid < ncand && cval[tid] >= kth_val;
    const unsigned long long bk = __ballot(in_k);
    if (lane == 0) s_redi[wave] = __popcll(bk);
    __syncthreads();
    int nk = 0;
    for (int w = 0; w < 16; ++w) nk += s_redi[w];
    __syncthreads();
    ncand = nk;
  }

  // ---- softmax numerators, inclusive scan ----------------------------------------------
  const float inv_t = 1.f / temp;
  const float zmax = cval[0];
  const float e = tid < ncand ? __expf((cval[tid] - zmax) * inv_t) : 0.f;
  float sc = e;                                              // inclusive scan within the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(sc, o, 64);
    if (lane >= o) sc += t;
  }
  if (lane == 63) wsum[wave] = sc;
  __syncthreads();
  float woff = 0.f, total = 0.f;
  for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; total += wsum[w]; }
  const float cum = sc + woff;                               // inclusive cumulative mass
  const float prev = cum - e;                                // cum[i-1]
  const bool keep = tid < ncand && (tid == 0 || top_p >= 1.f || !(prev / total > top_p));
  // kept entries form a prefix; its size and mass:
  const unsigned long long bal = __ballot(keep);
  if (lane == 0) s_redi[wave] = __popcll(bal);
  __syncthreads();
  int nkeep = 0;
  for (int w = 0; w < 16; ++w) nkeep += s_redi[w];
  __syncthreads();
  if (tid == nkeep - 1) s_red[0] = cum;
  __syncthreads();
  const float kept_total = s_red[0];

  // ---- categorical draw -------------------------------------------------------------------
  const float u = philox_uniform(p.seeds ? p.seeds[r] : 0ull, p.steps ? (unsigned)p.steps[r] : 0u);
  const float target = u * kept_total;
  const bool below = tid < nkeep && cum <= target;           // entries entirely below the target
  const unsigned long long bb = __ballot(below);
  __syncthreads();
  if (lane == 0) s_redi[wave] = __popcll(bb);
  __syncthreads();
  if (tid == 0) {
    int pick = 0;
    for (int w = 0; w < 16; ++w) pick += s_redi[w];
    if (pick >= nkeep) pick = nkeep - 1;
    emit_token(p, r, cidx[pick]);
  }
  if (p.dbg_probs && tid < p.dbg_cap) {
    p.dbg_probs[(long)r * p.dbg_cap + tid] = tid < nkeep ? e / kept_total : 0.f;
    p.dbg_ids[(long)r * p.dbg_cap + tid] = tid < nkeep ? cidx[tid] : -1;
  }
}

}  // namespace dlia

extern "C" void dlia_sample(int* out_tokens, const float* logits, const float* temperature,
                            const int* top_k, const float* top_p, const uint64_t* seeds,
                            const int* steps, int R, int vocab, long ldl, float* dbg_probs,
                            int* dbg_ids, int dbg_cap, hipStream_t stream, const float* chunk_max, long ldc,
                            int* ids_out, int* steps_adv, int* ctx_adv) {
  if (R <= 0) return;
  dlia::SampleParams p;
  const int nchunks = (vocab + dlia::kChunk - 1) / dlia::kChunk;
  const bool guided = chunk_max != nullptr && nchunks <= dlia::kMaxChunks && ldc >= nchunks;
  p.chunk_max = guided ? chunk_max : nullptr; p.ldc = ldc; p.nchunks = guided ? nchunks : 0;
  p.out = out_tokens; p.logits = logits; p.ldl = ldl; p.temperature = temperature; p.top_k = top_k;
  p.top_p = top_p; p.seeds = (const unsigned long long*)seeds; p.steps = steps; p.vocab = vocab;
  p.dbg_probs = dbg_probs; p.dbg_ids = dbg_ids; p.dbg_cap = dbg_cap > dlia::kCap ? dlia::kCap : dbg_cap;
  p.ids_out = ids_out; p.steps_adv = steps_adv; p.ctx_adv = ctx_adv;
  hipLaunchKernelGGL(dlia::sample_kernel, dim3(R), dim3(dlia::kSampThreads), 0, stream, p);
}
