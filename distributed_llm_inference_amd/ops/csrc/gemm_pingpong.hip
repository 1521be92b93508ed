// Compute-bound GEMM for more than 256 rows (large decode micro-batches, prefill chunks):
// out[M,N] = epi(x[M,K] @ W[N,K]^T), bf16, 256 x 256 output tile per workgroup, K walked in 64-wide
// tiles, both operands moved global -> LDS by DMA, MFMA 16x16x32.
//
// Structure ("ping-pong"): 8 waves, two per SIMD. The waves of a SIMD alternate roles every
// barrier interval: while one runs a cluster of 16 MFMAs (one quadrant of its 128 x 64 output for
// one K-tile), its partner issues the LDS fragment reads of ITS next quadrant and two DMA pieces
// of a future K-tile; at the next barrier they swap. The matrix pipe of a SIMD therefore sees one
// MFMA cluster after the other with no fragment read, DMA issue or wait in between, and every
// fragment read has a whole interval to land before its consumer starts. Waves 4-7 run one
// interval behind waves 0-3 (they take one extra barrier at the start, waves 0-3 one at the end).
//
// A K-tile is four 16 KiB half-tiles: A0 / A1 = x rows [0,128) / [128,256) of the block, B0 / B1 =
// weight columns [0,128) / [128,256). A wave owns 64 rows of EACH row half and 32 columns of EACH
// column half, so in every phase all eight waves work on the same pair of half-tiles:
//   q0: A0 x B0   (reads A0: 8 fragments, B0: 4)      q1: A0 x B1   (reads B1: 4)
//   q2: A1 x B1   (reads A1: 8)                        q3: A1 x B0   (B0 still in registers)
// 24 fragment reads per 64 MFMAs and wave (the 256 x 128-tile kernel of gemm_rows.hip: 20 per 32),
// half the DMA bytes per FLOP: less LDS and L2 energy per MFMA, which on this power-limited chip
// is clock, not just cycles.
//
// LDS: 2 K-tiles x 4 half-tiles = 128 KiB. The DMA stream is the sequence A0 B0 B1 A1 per K-tile
// (first-use order); phase P = 4 t + q issues half-tile number P + 6, two 1 KiB pieces per wave.
//  * write-after-read: the slot being refilled held the same half of K-tile (number - 8) / 4,
//    last read in phase number - 8 + {0,0,1,2}[kind] <= P - 2. Reads are issued in the first
//    interval of a phase and complete (lgkmcnt(0)) in its second, so two phases later every wave
//    of both groups has them in registers before any wave issues the DMA.
//  * read-after-write: at q3 every wave waits (counted vmcnt) until only the half-tile it issued in
//    q2 may still be in flight, i.e. its pieces of the NEXT K-tile have landed; the two barriers
//    between that wait and the next q0 reads (one for each group's wait) make that true for all
//    waves' pieces. Past the end of K the stream re-reads the last tile into slots nobody reads
//    again, so the count stays exact.
// Split-K (shapes with fewer output tiles than CUs, e.g. N = 4096 at 1024 rows = 64 tiles): the S
// workgroups of a tile run side by side on one XCD, each over 1/S of K, and the last of them to
// finish sums the tile (see the end of the kernel): S - 1 slabs of 256 KiB written and read once.
// Weights are the packed-B image (ops/packing.py): a (n-tile, k-step) block is 64 lanes x 16 B in
// fragment order, lane-linear in LDS, conflict-free; x rows are 128 B with the 16-B slot
// XOR-swizzled by (row >> 1) & 7 on the DMA source column and on the read.
//
// Replaces the nn.Linear calls inside HF LlamaDecoderLayer that the reference runs on CPU in fp32
// (Worker1.py:128-137) and the lm_head (orchestration.py:141).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gemm_common.h"
#include "launch.h"

// Diagnostic builds only (bench/pp_ablate.py): PP_ABL is a mask of parts of the main loop to leave
// out — 1 DMA issue, 2 fragment reads, 4 barriers, 8 MFMAs. Results are then wrong on purpose; what
// is measured is what the remaining parts cost. The library is always built with 0.
#ifndef PP_ABL
#define PP_ABL 0
#endif

namespace dlia {

constexpr int kPPThreads = 512;
constexpr int kPPHalf = 16384;                   // bytes of one half-tile
constexpr int kPPLds = 8 * kPPHalf;              // 2 K-tiles x {A0, B0, B1, A1}
constexpr int kPPSlab = 256 * 256;               // floats of one partial tile
constexpr int kPPCounters = 4096;                // counter words at the end of the workspace (two per output tile)

// Two 1 KiB pieces, global -> LDS, under ONE setting of M0: lane l of piece i lands at
// lds_addr + 1024 i + 16 l and reads g_i + 16 bytes per lane as set up by the caller. The second
// piece uses the instruction's immediate offset, which advances the global AND the LDS address, so
// its source pointer is passed 1024 bytes low. Why one setting: a write to M0 waits until every
// earlier LDS-DMA instruction of the wave has read it, ~180 cycles after issue — with M0 written
// (and restored) around every piece a wave issues one piece per ~180 cycles and a CU takes 32 bytes
// per cycle from L2, with four pieces per setting 63 (bench/l2_lds_bw.hip). M0 is not restored:
// nothing else in this kernel uses it (no other LDS-DMA, no indexed register access).
__device__ __forceinline__ void pp_dma_pair(const void* g0, const void* g1_minus_1k, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "global_load_lds_dwordx4 %2, off offset:1024"
               :: "s"(lds_addr), "v"(g0), "v"(g1_minus_1k) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void pp_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pp_loop_barrier() {
  if constexpr (!(PP_ABL & 4)) pp_barrier();
  else __builtin_amdgcn_sched_barrier(0);
}

template <int V> using ic = std::integral_constant<int, V>;
typedef int i32x8 __attribute__((ext_vector_type(8)));

// F8: both operands are OCP e4m3 bytes (x rows of p.ldx BYTES with per-row scales p.xscale, W from
// pack_weight_a8w8 with per-column scales p.wscale). Every byte count, LDS address, DMA piece and
// wait count is the same as for bf16: a 128-byte x row piece then holds 128 K-elements, the two
// 1 KiB weight blocks of a (n-tile, K-tile) are the two halves of one 32-byte-per-lane fragment, and a
// cluster is 8 x v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) instead of 16 x 16x16x32 —
// the same 512 matrix-pipe cycles, twice the arithmetic. Lane (r, q) of an x fragment holds
// k = 32 q .. 32 q + 31 of row r: the 16-byte slots 2 q and 2 q + 1 of its row.
// CMAX: the lm_head instantiation, whose epilogue also leaves the chunk maxima (GemmParams::chunk_max). It is an
// instantiation of its own because the extra epilogue code changes how hipcc allocates the whole kernel's registers
// (220 instead of 256 VGPRs, a different main loop): with it compiled into every GEMM, bench.py's default point lost
// 1.1 % (60.98 against 60.30 ms per step, four position-balanced runs each) — more than half of what the guided
// sampler it serves brings.
template <bool F8, bool CMAX = false>
__global__ __launch_bounds__(kPPThreads, 2) void gemm_pingpong_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;       // 64-row strip of each row half, 32-column strip of each column half

  // Workgroup ids go round-robin over the 8 XCDs. The row blocks of one column block run on one
  // XCD back to back, so its weight panel comes from HBM once and from that XCD's L2 afterwards.
  const int mblocks = (p.M + 255) >> 8;
  const int nb = (p.ntiles + 15) >> 4;           // 256-column blocks (the last may have one column half only)
  // ... and the K splits of one tile are neighbours in that order: dispatched together, done together.
  const int within = blockIdx.x >> 3;
  const int ks = within % p.splitk, rest = within / p.splitk;
  const int cb = (rest / mblocks) * 8 + (blockIdx.x & 7);
  if (cb >= nb) return;                          // grid padded to 8 column blocks; whole workgroup, before any barrier
  const int mblk = rest % mblocks, mrow0 = mblk * 256;
  const int tiles_total = p.K >> (F8 ? 7 : 6), KT = tiles_total * 2;   // K-tile = 128 bytes of an x row
  const int per = (tiles_total + p.splitk - 1) / p.splitk;
  const int c0 = ks * per;
  const int n = min(tiles_total, c0 + per) - c0;   // K-tiles of this split (>= 1 by the launcher)
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  // diagnostic mask 16: every workgroup of an XCD loads the same two panels (all L2 hits)
  const int cb_l = (PP_ABL & 16) ? (int)(blockIdx.x & 7) : cb, mrow0_l = (PP_ABL & 16) ? 0 : mrow0;
  // ---- DMA sources: per half-tile this wave moves pieces 2 wave and 2 wave + 1 -----------------
  // x piece: 8 rows x 128 B (rows past M re-read the last row; never stored). weight piece: one
  // (n-tile, k-step) block; a column half past the packed width re-reads n-tile `wave` (never stored).
  const long ldxb = F8 ? p.ldx : p.ldx * 2;      // row stride in bytes
  const char* asrc[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = 16 * wave + 8 * e + (lane >> 3);
      const int slot = (lane & 7) ^ ((r >> 1) & 7);
      asrc[h][e] = reinterpret_cast<const char*>(p.x) + (long)min(mrow0_l + h * 128 + r, p.M - 1) * ldxb + slot * 16 + (long)c0 * 128
                   - e * 1024;                     // the second piece of a pair is issued with offset:1024
    }
  const char* bsrc[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int nt = cb_l * 16 + h * 8 + wave;
    bsrc[h] = reinterpret_cast<const char*>(p.wp) + ((long)(nt < p.ntiles ? nt : wave) * KT + (long)c0 * 2) * 1024 + lane * 16;
  }
  // kind: 0 A0, 1 B0, 2 B1, 3 A1 (the order of the stream); c = K-tile of this split; par = its LDS buffer
  auto issue = [&](auto kind_c, int c, int par) {
    constexpr int kind = decltype(kind_c)::value;
    const int cc = min(c, n - 1);
    const unsigned dst = lds0 + (par * 4 + kind) * kPPHalf + wave * 2048;
    if constexpr (kind == 0 || kind == 3) {
      constexpr int h = kind == 3 ? 1 : 0;
      pp_dma_pair(asrc[h][0] + (long)cc * 128, asrc[h][1] + (long)cc * 128, dst);
    } else {
      constexpr int h = kind == 2 ? 1 : 0;
      const char* g = bsrc[h] + (long)cc * 2048;
      pp_dma_pair(g, g, dst);
    }
  };

  f32x4 acc[2][2][4][2];                         // [row half][column half][m-tile][n-tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 af[4][2], b0[2][2], b1[2][2];            // fragments: x of the current row half, weights of both column halves

  const int xoff = (wr * 64 + l15) * 128;
  const int sw = (l15 >> 1) & 7;
  const int boff = wc * 4096 + lane * 16;
  auto read_a = [&](int par, auto half_c) {
    constexpr int h = decltype(half_c)::value;
    const char* base = smem + (par * 4 + (h ? 3 : 0)) * kPPHalf + xoff;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        af[i][kk] = *reinterpret_cast<const u32x4*>(base + i * 2048 + (((F8 ? 2 * quad + kk : kk * 4 + quad) ^ sw) << 4));
  };
  auto read_b = [&](int par, auto half_c, u32x4 (&bf)[2][2]) {
    constexpr int h = decltype(half_c)::value;
    const char* base = smem + (par * 4 + (h ? 2 : 1)) * kPPHalf + boff;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) bf[j][kk] = *reinterpret_cast<const u32x4*>(base + (j * 2 + kk) * 1024);
  };
  auto cluster = [&](f32x4 (&c)[4][2], const u32x4 (&bf)[2][2]) {
    if constexpr (PP_ABL & 8) return;
    __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
      auto join = [](const u32x4& lo, const u32x4& hi) {
        return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      };
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          c[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(join(bf[j][0], bf[j][1]), join(af[i][0], af[i][1]), c[i][j],
                                                                     0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) c[i][j] = mfma16(bf[j][kk], af[i][kk], c[i][j]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // One phase = [fragment reads + DMA issue] barrier [MFMA cluster] barrier. t = K-tile, par = t & 1.
  unsigned stamp_sum[4][4] = {};                 // diagnostic (PP_ABL & 32): [phase kind][segment]
  unsigned long long stamp_prev = 0;
  auto phase = [&](auto q_c, int t, int par) {
    constexpr int q = decltype(q_c)::value;
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if constexpr (PP_ABL & 32) s0 = __builtin_amdgcn_s_memtime();
    if constexpr (q == 3) pp_wait_vm<2>();       // my pieces of K-tile t + 1 have landed (A0 of t + 2 may be in flight)
    if constexpr (!(PP_ABL & 2)) {
      if constexpr (q == 0) { read_b(par, ic<0>{}, b0); __builtin_amdgcn_sched_barrier(0); read_a(par, ic<0>{}); }
      if constexpr (q == 1) read_b(par, ic<1>{}, b1);
      if constexpr (q == 2) read_a(par, ic<1>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    // half-tile number 4 t + q + 6
    if constexpr (!(PP_ABL & 1)) {
      if constexpr (q == 0) issue(ic<2>{}, t + 1, par ^ 1);
      if constexpr (q == 1) issue(ic<3>{}, t + 1, par ^ 1);
      if constexpr (q == 2) issue(ic<0>{}, t + 2, par);
      if constexpr (q == 3) issue(ic<1>{}, t + 2, par);
    }
    if constexpr (PP_ABL & 32) s1 = __builtin_amdgcn_s_memtime();
    pp_loop_barrier();
    if constexpr (PP_ABL & 32) s2 = __builtin_amdgcn_s_memtime();
    if constexpr (q == 0) cluster(acc[0][0], b0);
    if constexpr (q == 1) cluster(acc[0][1], b1);
    if constexpr (q == 2) cluster(acc[1][1], b1);
    if constexpr (q == 3) cluster(acc[1][0], b0);
    if constexpr (PP_ABL & 32) {
      __builtin_amdgcn_sched_barrier(0);
      s3 = __builtin_amdgcn_s_memtime();
      // the previous phase's second barrier ended at this phase's s0
      if (stamp_prev) stamp_sum[(q + 3) & 3][3] += (unsigned)(s0 - stamp_prev);
      stamp_sum[q][0] += (unsigned)(s1 - s0);
      stamp_sum[q][1] += (unsigned)(s2 - s1);
      stamp_sum[q][2] += (unsigned)(s3 - s2);
      stamp_prev = s3;
      __builtin_amdgcn_sched_barrier(0);
    }
    pp_loop_barrier();
  };

  // prologue: K-tile 0 whole, A0 and B0 of K-tile 1
  issue(ic<0>{}, 0, 0); issue(ic<1>{}, 0, 0); issue(ic<2>{}, 0, 0); issue(ic<3>{}, 0, 0);
  issue(ic<0>{}, 1, 1); issue(ic<1>{}, 1, 1);
  pp_wait_vm<4>();                               // K-tile 0 has landed (my pieces)
  pp_barrier();                                  // ... and everyone's
  if constexpr (PP_ABL & 2) {                    // diagnostic: fragments read once, outside the loop
    read_b(0, ic<0>{}, b0); read_b(0, ic<1>{}, b1); read_a(0, ic<0>{});
  }
  if (wr == 1) pp_loop_barrier();                // waves 4-7 run one interval behind

  for (int t = 0; t < n; t += 2) {
    phase(ic<0>{}, t, 0); phase(ic<1>{}, t, 0); phase(ic<2>{}, t, 0); phase(ic<3>{}, t, 0);
    if (t + 1 < n) {
      phase(ic<0>{}, t + 1, 1); phase(ic<1>{}, t + 1, 1); phase(ic<2>{}, t + 1, 1); phase(ic<3>{}, t + 1, 1);
    }
  }
  if (wr == 0) pp_loop_barrier();                // the barrier waves 4-7 took at the start
  pp_wait_vm<0>();                               // no DMA may land after the workgroup is gone

  if constexpr (PP_ABL & 32) {
    if (lane == 0 && p.partial != nullptr) {
      unsigned* o = reinterpret_cast<unsigned*>(p.partial) + ((long)blockIdx.x * 8 + wave) * 16;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) o[a * 4 + b] = stamp_sum[a][b];
    }
  }
  if (p.splitk > 1) {
    // ---- sum the K splits of this tile ------------------------------------------------------
    // Each workgroup first draws a ticket from the tile's arrival counter. All but the last to
    // arrive store their accumulators (write-through, register order) to their slab and raise the
    // tile's second counter; the last one, which never stores, waits until that counter shows
    // every slab complete — a wait for workgroups that are already running and wait for nothing
    // themselves, so it cannot deadlock whatever else is resident — adds the slabs to its
    // registers and runs the fused epilogue.
    const int tile = cb * mblocks + mblk;
    int* const arrived = p.counters + 2 * tile;
    int* const stored = arrived + 1;
    float* const slabs = p.partial + (long)tile * (p.splitk - 1) * kPPSlab;
    int* const lds_word = reinterpret_cast<int*>(smem);      // the staging area is dead; one LDS object only
    __syncthreads();
    if (tid == 0) *lds_word = __hip_atomic_fetch_add(arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = *lds_word;
    if (ticket != p.splitk - 1) {
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(slabs + (long)ticket * kPPSlab, 0, kPPSlab * 4, 0x00020000);
#pragma unroll
      for (int f = 0; f < 32; ++f) {
        const f32x4 v = acc[f >> 4][(f >> 3) & 1][(f >> 1) & 3][f & 1];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (f * kPPThreads + tid) * 16, 0, 16);   // sc1
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains, then one lane signals
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(stored, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;                                                // whole workgroup
    }
    if (tid == 0) {
      while (__hip_atomic_load(stored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.splitk - 1)
        __builtin_amdgcn_s_sleep(8);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < p.splitk - 1; ++s) {
      const f32x4* src = reinterpret_cast<const f32x4*>(slabs + (long)s * kPPSlab) + tid;
#pragma unroll
      for (int f0 = 0; f0 < 32; f0 += 8) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(f0 + u) * kPPThreads];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + u;
          acc[f >> 4][(f >> 3) & 1][(f >> 1) & 3][f & 1] += t[u];
        }
      }
    }
    if (tid == 0) {                                          // both counters back to zero for the next launch
      __hip_atomic_store(arrived, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(stored, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if constexpr (F8) {
    // dequantise: per output column (weight row) and per row (token); sums over K splits are linear
#pragma unroll
    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int tile = min(cb * 16 + hn * 8 + wc * 2 + j, p.ntiles - 1);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.wscale + tile * 16 + quad * 4);
#pragma unroll
        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[hm][hn][i][j] *= sc;
      }
#pragma unroll
    for (int hm = 0; hm < 2; ++hm)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mrow0 + hm * 128 + wr * 64 + i * 16 + l15;
        const float xs = p.xscale[m < p.M ? m : p.M - 1];
#pragma unroll
        for (int hn = 0; hn < 2; ++hn)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[hm][hn][i][j] *= xs;
      }
  }
  if constexpr (CMAX) {
    // ---- lm_head: maximum of each 32-column chunk of each row ------------------------------------
    // A wave holds, per row half and column half, 64 rows x 32 columns: lane (l15, quad) has row
    // i * 16 + l15 and columns (j * 16 + quad * 4 + reg) of them. Eight values per lane, then a max over
    // the four quads (lanes 16 and 32 apart), and the quad whose number equals (i & 3) stores the value.
#pragma unroll
    for (int hm = 0; hm < 2; ++hm)
#pragma unroll
      for (int hn = 0; hn < 2; ++hn) {
        const int col0 = (cb * 16 + hn * 8 + wc * 2) * 16 + quad * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = -__builtin_inff();
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (col0 + j * 16 + r < p.n_valid) v = fmaxf(v, acc[hm][hn][i][j][r]);
          v = fmaxf(v, __shfl_xor(v, 16, 64));
          v = fmaxf(v, __shfl_xor(v, 32, 64));
          const int m = mrow0 + hm * 128 + wr * 64 + i * 16 + l15;
          if (quad == (i & 3) && m < p.M) p.chunk_max[(long)m * p.ldc + (cb * 8 + hn * 4 + wc)] = v;
        }
      }
  }
  // ---- row-wise epilogue for what a Llama layer asks of this kernel ------------------------------------
  // bf16 out, the whole 256-column block inside N: plain (+ bias, GELU, + residual: QKV, O, down, GPT-2's MLP) and SwiGLU (gate / up).
  // In the accumulator layout a lane holds 4 columns of one row of a tile: a store writes 8 bytes per lane into 16
  // different rows, 32 of those per wave (and as many residual loads). Here the block goes through the idle staging
  // area as fp32 — 16-byte slots XOR-ed with the row's low four bits, so the 16 rows a write instruction touches fall
  // on different banks — and every wave stores whole rows: 16 bytes per lane, two rows of 512 contiguous bytes (four
  // of 256 for SwiGLU) per instruction, the residual read the same way. Plain: one row half (128 x 256 fp32 = the
  // whole 128 KiB) at a time; SwiGLU: gate and up meet in registers, the 256 x 128 results are staged at once.
  if (!p.out_f32 && (cb + 1) * 256 <= p.N && (PP_ABL & 4096) == 0 &&
      (p.act != 1 || (p.residual == nullptr && p.bias == nullptr)) && ((p.ldo | p.ldr) & 7) == 0 &&
      ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.residual) | reinterpret_cast<uintptr_t>(p.bias)) & 15) == 0) {
    float* const st = reinterpret_cast<float*>(smem);
    if (p.act != 1) {
      const int cl = lane & 31, rsub = lane >> 5;
      float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (p.bias) unpack8(ld16(p.bias + cb * 256 + cl * 8), b8);
      bf16* const outp = reinterpret_cast<bf16*>(p.out) + cb * 256 + cl * 8;
      const bf16* const resp = p.residual + cb * 256 + cl * 8;     // (only dereferenced when there is a residual)
#pragma unroll
      for (int hm = 0; hm < 2; ++hm) {
        pp_barrier();                             // the area is free: all DMA landed and fragments read / previous half read back
#pragma unroll
        for (int hn = 0; hn < 2; ++hn)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              *reinterpret_cast<f32x4*>(st + (wr * 64 + i * 16 + l15) * 256 + (((hn * 32 + wc * 8 + j * 4 + quad) ^ l15) << 2)) =
                  acc[hm][hn][i][j];
        const int m0 = mrow0 + hm * 128 + wave * 16 + rsub;
        u32x4 res[8];
        if (p.residual) {
#pragma unroll
          for (int s = 0; s < 8; ++s) res[s] = ld16(resp + (long)min(m0 + 2 * s, p.M - 1) * p.ldr);
        }
        pp_barrier();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int R = wave * 16 + s * 2 + rsub;
          const float* row = st + R * 256;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(row + (((2 * cl) ^ (R & 15)) << 2));
          const f32x4 hi = *reinterpret_cast<const f32x4*>(row + (((2 * cl + 1) ^ (R & 15)) << 2));
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          if (p.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b8[e];
          }
          if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
          }
          if (p.residual) {
            float r8[8];
            unpack8(res[s], r8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r8[e];
          }
          if (m0 + 2 * s < p.M) st16(outp + (long)(m0 + 2 * s) * p.ldo, pack8(v));
        }
      }
    } else {
      pp_barrier();
#pragma unroll
      for (int hm = 0; hm < 2; ++hm)
#pragma unroll
        for (int hn = 0; hn < 2; ++hn)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = silu(acc[hm][hn][i][0][e]) * acc[hm][hn][i][1][e];
            *reinterpret_cast<f32x4*>(st + (hm * 128 + wr * 64 + i * 16 + l15) * 128 + (((hn * 16 + wc * 4 + quad) ^ l15) << 2)) = o;
          }
      const int cl = lane & 15, rsub = lane >> 4;
      bf16* const outp = reinterpret_cast<bf16*>(p.out) + cb * 128 + cl * 8;
      const int m0 = mrow0 + wave * 32 + rsub;
      pp_barrier();
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int R = wave * 32 + s * 4 + rsub;
        const float* row = st + R * 128;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(row + (((2 * cl) ^ (R & 15)) << 2));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(row + (((2 * cl + 1) ^ (R & 15)) << 2));
        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (m0 + 4 * s < p.M) st16(outp + (long)(m0 + 4 * s) * p.ldo, pack8(v));
      }
    }
    return;
  }
  if constexpr (PP_ABL & 2048) {                  // diagnostic: one quadrant of the epilogue only
    write_acc_tiles<4, 2, true>(p, acc[0][0], cb * 16 + wc * 2, l15, quad, 0, mrow0 + wr * 64);
    return;
  }
#pragma unroll
  for (int hm = 0; hm < 2; ++hm)
#pragma unroll
    for (int hn = 0; hn < 2; ++hn)
      write_acc_tiles<4, 2, true>(p, acc[hm][hn], cb * 16 + hn * 8 + wc * 2, l15, quad, 0, mrow0 + hm * 128 + wr * 64);
}


// ---------------------------------------------------------------------------------------------
// The same 256 x 256 tile with FOUR waves, one per SIMD, each owning 128 x 128 of it (64 rows of
// each row half x 64 columns of each column half): 256 accumulator registers per lane, and per
// K-tile 32 fragment reads for 128 MFMAs per wave instead of 24 for 64 — a third fewer LDS bytes
// per MFMA — and two barriers per 128 MFMAs instead of two per 16. No partner wave hides anything
// here: a phase first issues the LDS reads of a LATER phase's new operand, then its own 16 MFMAs,
// whose operands arrived during earlier phases.
//
// The unit is a K-slice of 32 (one MFMA k-step): four fragment sets (A0, A1, B0, B1: 16 registers
// each) and an order of quadrants that alternates between the even and the odd slice of a K-tile
// make "load one set while two others are multiplied" work out with no set loaded while in use:
//   slice kk = 0: A0B0 [load B1]  A0B1 [load A1]  A1B1 [load A0 of kk = 1]       A1B0 [load B1 of kk = 1]
//   slice kk = 1: A0B1 [load B0]  A0B0 [load A1]  A1B0 [load A0 of the next tile] A1B1 [load B0 of the next tile]
// Phase j = 4 kk + q of K-tile T: half-tiles of T + 1 are first read at j = 6 (A0), 7 (B0), 8 (B1), 9 (A1);
// those of T are last read at j = 2 (A0), 3 (B1), 4 (B0), 5 (A1), each consumed by the MFMAs one or
// two phases later. The DMA stream is A1 B1 of T + 1, A0 B0 of T + 2, one half-tile (four 1 KiB pieces
// per wave, one ahead of each group of four MFMAs) in each odd phase:
//  * write-after-read: every refill is issued at least two phases after the last read of its slot's
//    previous content, behind one of the two barriers of a K-tile (start of phases 5 and 7).
//  * read-after-write: before the barrier of phase 5 every wave waits until only the two half-tiles
//    issued last may be in flight (A0, B0 of T + 1 have landed), before the one of phase 7 until only
//    one may be (B1, A1 of T + 1 have landed).
// bf16, no K split. Measured against the eight-wave kernel above (same box, 4096 rows, Llama-3-8B shapes): down
// projection (K 14336, one round of 256 workgroups) 1.49 against 1.45-1.47 PFLOP/s, gate/up (K 4096, seven rounds)
// 1.29-1.30 against 1.39-1.41: the main loop is a little faster, the serial prologue / epilogue of a workgroup whose
// four waves each own 64 output tiles costs more than that at short K. Ablations: without barriers and landing waits
// +2.4 %, without the DMA +18-20 % — the global -> LDS traffic, not the fragment reads or the synchronisation, is what
// both kernels pay for. The eight-wave kernel stays the default; DLIA_GEMM_LARGE=quad selects this one.
constexpr int kQThreads = 256;

// One 1 KiB piece global -> LDS in the scalar-base form: address = base (SGPR pair, wave-uniform) + a 32-bit lane
// offset + the instruction's immediate, which also advances the LDS address from M0. A lone wave per SIMD issues its
// own DMA between its MFMAs; with a scalar base the per-tile address arithmetic is scalar, no vector instruction is
// spent on 64-bit lane addresses (measured: same time as with per-lane pointers — it is the transfers that cost,
// not their issue). FIRST: sets M0 for the four pieces of a half-tile.
template <int E, bool FIRST>
__device__ __forceinline__ void q_dma_piece(const void* sbase, unsigned voff, unsigned lds_addr) {
  if constexpr (FIRST)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
                 :: "s"(lds_addr), "v"(voff), "s"(sbase), "n"(E * 1024) : "memory", "m0");
  else
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" :: "v"(voff), "s"(sbase), "n"(E * 1024) : "memory");
}

// The accumulators are pinned to the accumulation registers ("a" constraint): left to itself hipcc keeps part of the
// 256 in ordinary registers, runs out of those, and shuttles values between the two files around every MFMA (600
// copies and 90 spills per K-tile pair). Within the loop no MFMA reads an accumulator that one of the previous 100+
// MFMAs wrote, so no wait states are owed between them; the epilogue waits explicitly before it reads them.
__device__ __forceinline__ void mfma16_acc(f32x4& c, const u32x4& a, const u32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// The first product into an accumulator (C = 0): the accumulators are never initialised through ordinary registers.
__device__ __forceinline__ void mfma16_acc_first(f32x4& c, const u32x4& a, const u32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

// Accumulator -> ordinary registers, one tile at a time and in program order (volatile): left to the compiler, all 256
// are copied out at once after the loop, and the 256 ordinary registers that takes push everything else into scratch.
__device__ __forceinline__ f32x4 acc_read(f32x4 c) {
  float x0, x1, x2, x3;
  asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
               : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "a"(c[0]), "a"(c[1]), "a"(c[2]), "a"(c[3]));
  return f32x4{x0, x1, x2, x3};
}

__global__ __launch_bounds__(kQThreads, 1) void gemm_quad_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long st0 = 0, st1 = 0, st2 = 0;     // diagnostic (PP_ABL & 1024): kernel start, first MFMA, end of the K loop
  if constexpr (PP_ABL & 1024) st0 = __builtin_amdgcn_s_memtime();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  const int mblocks = (p.M + 255) >> 8;
  const int nb = (p.ntiles + 15) >> 4;
  const int within = blockIdx.x >> 3;
  const int cb = (within / mblocks) * 8 + (blockIdx.x & 7);
  if (cb >= nb) return;
  const int mblk = within % mblocks, mrow0 = mblk * 256;
  const int n = p.K >> 6, KT = n * 2;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  // ---- DMA sources: per half-tile this wave moves pieces 4 wave .. 4 wave + 3 --------------------
  // x piece e: 8 rows x 128 B (rows past M re-read the last row; never stored); weight piece e: (n-tile 2 wave + e / 2,
  // k-step e % 2), a column half past the packed width re-reads the first one. Lane offsets are relative to a base
  // that sits 3 KiB low, so that "offset - e KiB" (the immediate adds e KiB back) is never negative.
  const long ldxb = p.ldx * 2;
  const char* const abase = reinterpret_cast<const char*>(p.x) + (long)mrow0 * ldxb - 3072;
  const char* const bbase = reinterpret_cast<const char*>(p.wp) + (long)cb * 16 * KT * 1024 - 3072;
  unsigned aoff[2][4], boff4[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 32 * wave + 8 * e + (lane >> 3);
      const int slot = (lane & 7) ^ ((r >> 1) & 7);
      aoff[h][e] = (unsigned)((long)(min(mrow0 + h * 128 + r, p.M - 1) - mrow0) * ldxb) + slot * 16 + 3072 - e * 1024;
      const int j = 2 * wave + (e >> 1);
      const int nt = (cb * 16 + h * 8 + j < p.ntiles) ? h * 8 + j : j;
      boff4[h][e] = (unsigned)(nt * KT + (e & 1)) * 1024u + lane * 16 + 3072 - e * 1024;
    }
  // kind: 0 A0, 1 B0, 2 B1, 3 A1 (the LDS slot within a K-tile's buffer); c = K-tile (its buffer: c & 1); piece e
  auto issue_piece = [&](auto kind_c, int c, auto e_c) {
    constexpr int kind = decltype(kind_c)::value, e = decltype(e_c)::value;
    const int cc = min(c, n - 1);
    const unsigned dst = lds0 + ((c & 1) * 4 + kind) * kPPHalf + wave * 4096;
    if constexpr (kind == 0 || kind == 3) q_dma_piece<e, e == 0>(abase + (long)cc * 128, aoff[kind == 3 ? 1 : 0][e], dst);
    else                                  q_dma_piece<e, e == 0>(bbase + (long)cc * 2048, boff4[kind == 2 ? 1 : 0][e], dst);
  };
  auto issue = [&](auto kind_c, int c) {
    issue_piece(kind_c, c, ic<0>{}); issue_piece(kind_c, c, ic<1>{}); issue_piece(kind_c, c, ic<2>{}); issue_piece(kind_c, c, ic<3>{});
  };

  f32x4 acc[2][2][4][4];                         // [row half][column half][m-tile][n-tile]; first written by K-slice 0 of K-tile 0
  u32x4 a0[4], a1[4], b0[4], b1[4];              // fragments of one K-slice

  const int xoff = (wr * 64 + l15) * 128;
  const int sw = (l15 >> 1) & 7;
  const int boff = wc * 8192 + lane * 16;
  // (par, h, kk) -> fragment g of a set: row half h of the x tile / column half h of the weight tile, K-slice kk
  auto read_a1 = [&](int par, auto half_c, auto kk_c, u32x4 (&af)[4], int g) {
    constexpr int h = decltype(half_c)::value, kk = decltype(kk_c)::value;
    af[g] = *reinterpret_cast<const u32x4*>(smem + (par * 4 + (h ? 3 : 0)) * kPPHalf + xoff + (((kk * 4 + quad) ^ sw) << 4) + g * 2048);
  };
  auto read_b1 = [&](int par, auto half_c, auto kk_c, u32x4 (&bf)[4], int g) {
    constexpr int h = decltype(half_c)::value, kk = decltype(kk_c)::value;
    bf[g] = *reinterpret_cast<const u32x4*>(smem + (par * 4 + (h ? 2 : 1)) * kPPHalf + boff + kk * 1024 + g * 2048);
  };
  // Group g of a phase's 16 MFMAs: row g of the quadrant (BYCOL = false) or column g (true) — whichever makes group g
  // need only fragment g of the set that the previous phase loaded (an A set: rows; a B set: columns), so that a read
  // issued before group g of one phase is first needed by group g of the next, a whole phase later.
  auto mfma_group = [&](f32x4 (&c)[4][4], const u32x4 (&af)[4], const u32x4 (&bf)[4], int g, auto bycol_c, auto first_c) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = decltype(bycol_c)::value ? u : g, j = decltype(bycol_c)::value ? g : u;
      if constexpr (decltype(first_c)::value) mfma16_acc_first(c[i][j], bf[j], af[i]);
      else mfma16_acc(c[i][j], bf[j], af[i]);
    }
  };

  // Phase (kk, q) of K-tile t (buffer par = t & 1, a compile-time constant at every call site): four groups of
  // [one LDS read for a later phase, four MFMAs] — a lone wave pays for a burst of reads ahead of its MFMAs, reads
  // spread between them are almost free.
  auto phase = [&](auto par_c, auto kk_c, auto q_c, int t, auto first_c) {
    constexpr int par = decltype(par_c)::value, kk = decltype(kk_c)::value, q = decltype(q_c)::value;
    constexpr bool first = decltype(first_c)::value && kk == 0;       // K-slice 0 of K-tile 0
    constexpr int j = 4 * kk + q;
    auto rd = [&](int g) {
      if constexpr (kk == 0) {
        if constexpr (q == 0) read_b1(par, ic<1>{}, ic<0>{}, b1, g);
        if constexpr (q == 1) read_a1(par, ic<1>{}, ic<0>{}, a1, g);
        if constexpr (q == 2) read_a1(par, ic<0>{}, ic<1>{}, a0, g);
        if constexpr (q == 3) read_b1(par, ic<1>{}, ic<1>{}, b1, g);
      } else {
        if constexpr (q == 0) read_b1(par, ic<0>{}, ic<1>{}, b0, g);
        if constexpr (q == 1) read_a1(par, ic<1>{}, ic<1>{}, a1, g);
        if constexpr (q == 2) read_a1(par ^ 1, ic<0>{}, ic<0>{}, a0, g);
        if constexpr (q == 3) read_b1(par ^ 1, ic<0>{}, ic<0>{}, b0, g);
      }
    };
    auto mm = [&](int g) {
      using F = std::integral_constant<bool, first>;
      using C = std::integral_constant<bool, (q < 2)>;      // phases 0, 1 follow the load of a B set, phase 2 that of an A set
      if constexpr (kk == 0) {
        if constexpr (q == 0) mfma_group(acc[0][0], a0, b0, g, C{}, F{});
        if constexpr (q == 1) mfma_group(acc[0][1], a0, b1, g, C{}, F{});
        if constexpr (q == 2) mfma_group(acc[1][1], a1, b1, g, C{}, F{});
        if constexpr (q == 3) mfma_group(acc[1][0], a1, b0, g, C{}, F{});
      } else {
        if constexpr (q == 0) mfma_group(acc[0][1], a0, b1, g, C{}, std::false_type{});
        if constexpr (q == 1) mfma_group(acc[0][0], a0, b0, g, C{}, std::false_type{});
        if constexpr (q == 2) mfma_group(acc[1][0], a1, b0, g, C{}, std::false_type{});
        if constexpr (q == 3) mfma_group(acc[1][1], a1, b1, g, C{}, std::false_type{});
      }
    };
    auto dma = [&](auto e_c) {
      if constexpr ((j & 1) && !(PP_ABL & 256)) {
        // stream: ... A1 B1 of t + 1 (phases 1, 3), A0 B0 of t + 2 (phases 5, 7), one piece ahead of each MFMA group
        if constexpr (j == 1) issue_piece(ic<3>{}, t + 1, e_c);
        if constexpr (j == 3) issue_piece(ic<2>{}, t + 1, e_c);
        if constexpr (j == 5) issue_piece(ic<0>{}, t + 2, e_c);
        if constexpr (j == 7) issue_piece(ic<1>{}, t + 2, e_c);
      }
    };
    if constexpr (j == 5) { if constexpr (!(PP_ABL & 64)) pp_wait_vm<8>(); if constexpr (!(PP_ABL & 128)) pp_barrier(); }   // A0, B0 of t + 1 have landed; their readers of t are done
    if constexpr (j == 7) { if constexpr (!(PP_ABL & 64)) pp_wait_vm<4>(); if constexpr (!(PP_ABL & 128)) pp_barrier(); }   // B1, A1 of t + 1 have landed
    rd(0); dma(ic<0>{}); __builtin_amdgcn_sched_barrier(0); mm(0); __builtin_amdgcn_sched_barrier(0);
    rd(1); dma(ic<1>{}); __builtin_amdgcn_sched_barrier(0); mm(1); __builtin_amdgcn_sched_barrier(0);
    rd(2); dma(ic<2>{}); __builtin_amdgcn_sched_barrier(0); mm(2); __builtin_amdgcn_sched_barrier(0);
    rd(3); dma(ic<3>{}); __builtin_amdgcn_sched_barrier(0); mm(3); __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: K-tile 0 whole, A0 and B0 of K-tile 1 — the state the stream is in at the start of every K-tile
  issue(ic<0>{}, 0); issue(ic<1>{}, 0); issue(ic<3>{}, 0); issue(ic<2>{}, 0);
  issue(ic<0>{}, 1); issue(ic<1>{}, 1);
  pp_wait_vm<8>();                                // K-tile 0 has landed (my pieces)
  pp_barrier();                                   // ... and everyone's
#pragma unroll
  for (int g = 0; g < 4; ++g) { read_a1(0, ic<0>{}, ic<0>{}, a0, g); read_b1(0, ic<0>{}, ic<0>{}, b0, g); }

  auto tile = [&](auto par_c, int t, auto f) {
    phase(par_c, ic<0>{}, ic<0>{}, t, f); phase(par_c, ic<0>{}, ic<1>{}, t, f); phase(par_c, ic<0>{}, ic<2>{}, t, f); phase(par_c, ic<0>{}, ic<3>{}, t, f);
    phase(par_c, ic<1>{}, ic<0>{}, t, f); phase(par_c, ic<1>{}, ic<1>{}, t, f); phase(par_c, ic<1>{}, ic<2>{}, t, f); phase(par_c, ic<1>{}, ic<3>{}, t, f);
  };
  if constexpr (PP_ABL & 1024) st1 = __builtin_amdgcn_s_memtime();
  tile(ic<0>{}, 0, std::true_type{});
  for (int t = 1; t < n; t += 2) {
    tile(ic<1>{}, t, std::false_type{});
    if (t + 1 < n) tile(ic<0>{}, t + 1, std::false_type{});
  }
  if constexpr (PP_ABL & 1024) st2 = __builtin_amdgcn_s_memtime();
  pp_wait_vm<0>();                                // no DMA may land after the workgroup is gone
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs have written their accumulators
  __builtin_amdgcn_sched_barrier(0);

  // Epilogue through LDS, as a LOOP over tiles: a wave's 64 tiles unrolled through the general epilogue are ~9,400
  // instructions of straight-line code that every workgroup runs once — instruction fetch, not arithmetic or stores,
  // made that epilogue take ~19 us per workgroup (three times the eight-wave kernel's, whose waves run half as much
  // code each). Per quadrant a wave parks its 16 tiles in its own 16 KiB of the (now idle) staging area, lane-linear,
  // and a rolled loop reads one tile (or a gate / up pair) back and finishes it: the epilogue's code exists once.
  pp_barrier();                                   // every wave has read its last fragments: the staging area is free
  unsigned long long st2b = 0;
  if constexpr (PP_ABL & 1024) st2b = __builtin_amdgcn_s_memtime();
  char* const stage = smem + wave * 17408 + lane * 16;      // (each wave has 17408 bytes: the row-wise fast paths pad their rows)
  auto store_rows = [&](auto hm_c, auto hn_c, auto i_c) {       // (diagnostic builds only: one row of tiles)
    constexpr int hm = decltype(hm_c)::value, hn = decltype(hn_c)::value, i = decltype(i_c)::value;
    f32x4 row[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) row[0][j] = acc_read(acc[hm][hn][i][j]);
    write_acc_tiles<1, 4, true>(p, row, cb * 16 + hn * 8 + wc * 4, l15, quad, 0, mrow0 + hm * 128 + wr * 64 + i * 16);
  };
  // Fast paths for what a Llama layer asks of this kernel — bf16 out, no bias, whole quadrant inside N: plain or
  // + residual (QKV, O, down) and SwiGLU (gate / up). In the accumulator layout a lane holds 4 columns of one row of a
  // tile, so a store instruction writes 8 bytes per lane into 16 different rows (32-byte pieces of 16 cache lines), and
  // 64 of those per wave, ~250 cycles apart, were the 16-20 thousand cycles a workgroup spent after its last MFMA. The
  // quadrant goes through the wave's own LDS area instead (fp32, rows padded to 272 / 144 bytes) and comes back row-wise:
  // 16 bytes per lane, 8 (16) whole rows of 128 (64) contiguous bytes per store, the residual read the same way.
  constexpr int kRowF = 68, kRowH = 36;           // floats per staged row: 64 (32) + 4 of padding
  float* const stq = reinterpret_cast<float*>(smem + wave * 17408);
  auto fast_plain = [&](auto hm_c, auto hn_c, auto res_c) __attribute__((always_inline)) {
    constexpr int hm = decltype(hm_c)::value, hn = decltype(hn_c)::value;
    constexpr bool RES = decltype(res_c)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(stq + (i * 16 + l15) * kRowF + j * 16 + quad * 4) = acc_read(acc[hm][hn][i][j]);
    const int col0 = (cb * 16 + hn * 8 + wc * 4) * 16 + (lane & 7) * 8;
    const int row0 = mrow0 + hm * 128 + wr * 64 + (lane >> 3);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int m = row0 + k * 8, mc = min(m, p.M - 1);
      const float* src = stq + (k * 8 + (lane >> 3)) * kRowF + (lane & 7) * 8;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if constexpr (RES) {
        float r[8];
        unpack8(ld16(p.residual + (long)mc * p.ldr + col0), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
      }
      if (m < p.M) st16(reinterpret_cast<bf16*>(p.out) + (long)mc * p.ldo + col0, pack8(v));
    }
  };
  auto fast_swiglu = [&](auto hm_c, auto hn_c) __attribute__((always_inline)) {
    constexpr int hm = decltype(hm_c)::value, hn = decltype(hn_c)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {            // (gate tile, up tile) -> one output tile
        const f32x4 g = acc_read(acc[hm][hn][i][2 * jp]), u = acc_read(acc[hm][hn][i][2 * jp + 1]);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = silu(g[e]) * u[e];
        *reinterpret_cast<f32x4*>(stq + (i * 16 + l15) * kRowH + jp * 16 + quad * 4) = o;
      }
    const int col0 = ((cb * 16 + hn * 8 + wc * 4) >> 1) * 16 + (lane & 3) * 8;
    const int row0 = mrow0 + hm * 128 + wr * 64 + (lane >> 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int m = row0 + k * 16;
      const float* src = stq + (k * 16 + (lane >> 2)) * kRowH + (lane & 3) * 8;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
      const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (m < p.M) st16(reinterpret_cast<bf16*>(p.out) + (long)m * p.ldo + col0, pack8(v));
    }
  };
  auto store_quadrant = [&](auto hm_c, auto hn_c) __attribute__((always_inline)) {
    constexpr int hm = decltype(hm_c)::value, hn = decltype(hn_c)::value;
    const int tile_base = cb * 16 + hn * 8 + wc * 4, row_base = mrow0 + hm * 128 + wr * 64 + l15;
    if (!p.out_f32 && p.bias == nullptr && (PP_ABL & 4096) == 0 && ((p.ldo | p.ldr) & 7) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.residual)) & 15) == 0) {
      if (p.act == 0 && (tile_base + 4) * 16 <= p.N) {
        if (p.residual) fast_plain(hm_c, hn_c, std::true_type{});
        else            fast_plain(hm_c, hn_c, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        return;
      }
      if (p.act == 1 && p.residual == nullptr && ((tile_base >> 1) + 2) * 16 <= (p.N >> 1)) {
        fast_swiglu(hm_c, hn_c);
        __builtin_amdgcn_sched_barrier(0);
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(stage + (i * 4 + j) * 1024) = acc_read(acc[hm][hn][i][j]);
    if (p.act == 1) {
#pragma unroll 2
      for (int t = 0; t < 8; ++t) {               // (gate tile, up tile) pairs
        const int i = t >> 1, j = (t & 1) * 2;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(stage + (i * 4 + j) * 1024);
        const f32x4 uv = *reinterpret_cast<const f32x4*>(stage + (i * 4 + j + 1) * 1024);
        const int m = row_base + i * 16, n = ((tile_base + j) >> 1) * 16 + quad * 4;
        if (m < p.M && n < (p.N >> 1)) {
          float g[4] = {gv[0], gv[1], gv[2], gv[3]};
          const float u[4] = {uv[0], uv[1], uv[2], uv[3]};
          epilogue4(p, m, n, g, u);
        }
      }
    } else {
#pragma unroll 4
      for (int t = 0; t < 16; ++t) {
        const int i = t >> 2, j = t & 3;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(stage + t * 1024);
        const int m = row_base + i * 16, n = (tile_base + j) * 16 + quad * 4;
        if (m < p.M && n < p.N) {
          float g[4] = {gv[0], gv[1], gv[2], gv[3]};
          epilogue4(p, m, n, g, g);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  if constexpr (PP_ABL & 512) {                   // diagnostic: one row of tiles only (what the rest of the epilogue costs)
    store_rows(ic<0>{}, ic<0>{}, ic<0>{});
    return;
  }
  store_quadrant(ic<0>{}, ic<0>{}); store_quadrant(ic<0>{}, ic<1>{}); store_quadrant(ic<1>{}, ic<0>{}); store_quadrant(ic<1>{}, ic<1>{});
  if constexpr (PP_ABL & 1024) {
    const unsigned long long st3a = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && p.partial != nullptr) {       // (bench/pp_ablate.py passes a stamp buffer as the workspace)
      unsigned* o = reinterpret_cast<unsigned*>(p.partial) + ((long)blockIdx.x * 8 + wave) * 16;
      o[0] = (unsigned)(st1 - st0); o[1] = (unsigned)(st2 - st1); o[2] = (unsigned)(st3 - st2); o[3] = 1u;
      o[4] = (unsigned)(st2b - st2); o[5] = (unsigned)(st3a - st2b); o[6] = (unsigned)(st3 - st3a);
    }
  }
}

}  // namespace dlia

using namespace dlia;

// Which kernel serves bf16 GEMMs of more than 256 rows: 0 = this one (default), 2 = the 256 x 128
// loader-wave kernel of gemm_rows.hip (also: DLIA_GEMM_LARGE=stream), 4 = the four-wave kernel above for
// launches that do not split K (DLIA_GEMM_LARGE=quad). which < 0 only reads.
extern "C" int dlia_gemm_large_choice(int which) {
  static std::atomic<int> v{[] { const char* e = getenv("DLIA_GEMM_LARGE"); return (e && e[0] == 's') ? 2 : ((e && e[0] == 'q') ? 4 : 0); }()};
  return which < 0 ? v.load() : v.exchange(which);
}

// K splits for a shape: as many as keep the launch within `target` workgroups, each split at least
// 8 K-tiles long, at most 8 splits. Default target 128, half the CUs: decode runs two micro-batches
// on two streams, and a thin launch shares the chip with the other stream's kernels. A caller that
// runs alone (a prefill chunk) asks for 256: 2048-token Llama-3-8B prefill 28.8 -> 25.4 ms. (Also
// splitting the 96-tile QKV shape in two at 1024 rows: 33.3 against 32.95 ms per decode step, not kept.)
static int pp_splitk(int M, int N, int ktiles, int want = 0) {
  static const int dflt = [] { const char* e = getenv("DLIA_GEMM_PP_TARGET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 128; }();
  const int target = want > 1 ? want : dflt;
  const int tiles = ((M + 255) / 256) * ((packed_ntiles(N) + 15) / 16);
  if (tiles > kPPCounters / 2) return 1;
  int s = target / tiles;
  if (s > 8) s = 8;
  while (s > 1 && ktiles / s < 8) --s;
  return s < 1 ? 1 : s;
}
extern "C" int dlia_gemm_pingpong_splitk(int M, int N, int K) { return pp_splitk(M, N, K / 64); }
// Workspace the split needs: slabs [tiles][S][256 x 256]; the arrival counters are the LAST
// kPPCounters words of whatever workspace the caller hands over (so that workspaces shared with
// other users of their front part keep them intact).
extern "C" long dlia_gemm_pingpong_ws_floats(int M, int N, int K) {
  const long tiles = (long)((M + 255) / 256) * ((packed_ntiles(N) + 15) / 16);
  const int s = std::max(pp_splitk(M, N, K / 64, 0), pp_splitk(M, N, K / 64, 256));   // whichever target a caller asks for
  return s > 1 ? tiles * (s - 1) * kPPSlab + kPPCounters : 0;
}

// bf16, M > 256 (any M is correct; below that the weight-streaming kernels are faster), K a
// multiple of 64; `p` fully set up by the caller. p->splitk > 1 needs p->partial = slabs for
// [tiles][splitk - 1] and p->counters = two zero words per tile (the kernel leaves them zero).
template <bool F8, bool CMAX = false>
static void launch_pingpong(const dlia::GemmParams* p, hipStream_t stream) {
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pingpong_kernel<F8, CMAX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kPPLds);
    return true;
  }();
  (void)once;
  const int nb = (p->ntiles + 15) / 16, nb8 = (nb + 7) / 8 * 8, mblocks = (p->M + 255) / 256;
  hipLaunchKernelGGL((gemm_pingpong_kernel<F8, CMAX>), dim3(nb8 * mblocks * p->splitk), dim3(kPPThreads), kPPLds, stream, *p);
}
static void launch_quad(const dlia::GemmParams* p, hipStream_t stream) {
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_quad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPPLds);
    return true;
  }();
  (void)once;
  const int nb = (p->ntiles + 15) / 16, nb8 = (nb + 7) / 8 * 8, mblocks = (p->M + 255) / 256;
  hipLaunchKernelGGL(gemm_quad_kernel, dim3(nb8 * mblocks), dim3(kQThreads), kPPLds, stream, *p);
}
extern "C" void dlia_gemm_pingpong(const dlia::GemmParams* p, hipStream_t stream) {
  if (p->chunk_max != nullptr) launch_pingpong<false, true>(p, stream);
  else if (dlia_gemm_large_choice(-1) == 4 && p->splitk == 1 && p->K >= 128) launch_quad(p, stream);
  else                         launch_pingpong<false>(p, stream);
}

// out[M,N] = epi(x[M,K] @ Wp^T) for bf16 weights and more than 256 rows (epilogues 0..5 as
// dlia_gemm_skinny). `chunk_max` (may be null; then ldc / n_valid are ignored): see GemmParams. `workspace` (may be null) of `ws_floats` floats: lets thin shapes split K; its
// last 4096 words are the arrival counters (zero when first handed over, zero after every launch).
extern "C" void dlia_gemm_large(void* out, const void* x, const void* wp, const void* bias, const void* residual,
                                float* workspace, long ws_floats, int M, int N, int K, long ldx, long ldo, long ldr,
                                int epi, int out_f32, int wg_target, hipStream_t stream, float* chunk_max, long ldc,
                                int n_valid) {
  if (M <= 0 || N <= 0) return;
  GemmParams p;
  p.chunk_max = chunk_max; p.ldc = ldc; p.n_valid = n_valid;
  p.x = (const bf16*)x; p.ldx = ldx; p.wp = (const u32x4*)wp; p.wscale = nullptr; p.xscale = nullptr; p.out = out; p.ldo = ldo;
  p.bias = (epi == 1 || epi == 3 || epi == 5) ? (const bf16*)bias : nullptr;
  p.residual = (epi == 2 || epi == 3) ? (const bf16*)residual : nullptr;
  p.ldr = ldr; p.partial = workspace; p.M = M; p.N = N; p.K = K;
  p.ntiles = packed_ntiles(N);
  p.act = epi == 4 ? 1 : (epi == 5 ? 2 : 0);
  p.out_f32 = out_f32; p.partial_only = 0; p.splitk = 1; p.group_n = 1; p.counters = nullptr;
  if (dlia_gemm_large_choice(-1) == 2 && chunk_max == nullptr) { dlia_gemm_rows(&p, stream); return; }
  const int s = pp_splitk(M, N, K / 64, wg_target);
  const long tiles = (long)((M + 255) / 256) * ((p.ntiles + 15) / 16);
  if (workspace && s > 1 && ws_floats >= tiles * (s - 1) * kPPSlab + kPPCounters) {
    p.splitk = s;
    p.counters = reinterpret_cast<int*>(workspace + ws_floats - kPPCounters);
  }
  dlia_gemm_pingpong(&p, stream);
}

// The same for e4m3 x e4m3 (dlia_gemm_a8w8's contract: xq bytes [M, K] with row stride ldx BYTES,
// xscale[M], wp from pack_weight_a8w8, wscale per packed row), more than 256 rows, K a multiple of 128.
extern "C" void dlia_gemm_a8w8_large(void* out, const void* xq, const float* xscale, const void* wp, const float* wscale,
                                     const void* bias, const void* residual, float* workspace, long ws_floats, int M,
                                     int N, int K, long ldx, long ldo, long ldr, int epi, int out_f32, int wg_target,
                                     hipStream_t stream) {
  if (M <= 0 || N <= 0) return;
  GemmParams p;
  p.x = (const bf16*)xq; p.ldx = ldx; p.wp = (const u32x4*)wp; p.wscale = wscale; p.xscale = xscale; p.out = out; p.ldo = ldo;
  p.bias = (epi == 1 || epi == 3 || epi == 5) ? (const bf16*)bias : nullptr;
  p.residual = (epi == 2 || epi == 3) ? (const bf16*)residual : nullptr;
  p.ldr = ldr; p.partial = workspace; p.M = M; p.N = N; p.K = K;
  p.ntiles = packed_ntiles(N);
  p.act = epi == 4 ? 1 : (epi == 5 ? 2 : 0);
  p.out_f32 = out_f32; p.partial_only = 0; p.splitk = 1; p.group_n = 1; p.counters = nullptr;
  if (dlia_gemm_large_choice(-1) == 2) { p.partial = nullptr; dlia_gemm_rows_f8(&p, stream); return; }
  const int s = pp_splitk(M, N, K / 128, wg_target);
  const long tiles = (long)((M + 255) / 256) * ((p.ntiles + 15) / 16);
  if (workspace && s > 1 && ws_floats >= tiles * (s - 1) * kPPSlab + kPPCounters) {
    p.splitk = s;
    p.counters = reinterpret_cast<int*>(workspace + ws_floats - kPPCounters);
  }
  launch_pingpong<true>(&p, stream);
}
