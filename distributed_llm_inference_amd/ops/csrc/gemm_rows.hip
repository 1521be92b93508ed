// Weight-streaming GEMM (bf16, and e4m3 x e4m3: see F8 below) with both operands moved by DMA:
// out[M,N] = epi(x[M,K] @ W[N,K]^T) for M > 128 — decode batches of 129..256 rows (one row block, split-K partials available) and,
// above that, prefill chunks / large decode batches as a grid of 256-row blocks.
//
// A workgroup owns 256 x 128 outputs and walks K in 64-wide chunks. Every workgroup re-stages its
// x rows from L2 while it streams its own weight columns from HBM; measured on MI355X with the
// compute removed (256 rows, Llama-3-8B gate/up shape), x alone moves at the L2 ceiling
// (19 TB/s), the weights alone at the HBM ceiling (5.5 TB/s), both together in ~55 us: the
// kernel's job is to add as little as possible to that. It therefore
//  * moves BOTH operands global -> LDS by DMA (no register ring): three 32 KiB stages of x
//    (prefetched two chunks ahead) and a ring of three 16 KiB weight slots — 144 KiB of LDS, which
//    leaves room for a small kernel of another stream on the CU — with one hand-counted vmcnt +
//    one barrier per chunk; the weight DMA is non-temporal so the stream cannot evict x from L2;
//  * splits the workgroup into 8 compute waves (two per SIMD) — wave (mh, nq) owns rows
//    [128 mh, +128) x n-tiles [2 nq, +2), 64 accumulator registers, the two row halves sharing
//    each weight fragment through the LDS ring — and 4 loader waves that do nothing but issue the
//    DMA: with the loads issued by the compute waves themselves the same kernel took 72 us
//    instead of 59 (a wave blocked in a full memory queue issues no MFMAs).
// The kernel is written for any ROWS in {32, 64, 128, 256} per workgroup (fewer rows shrink the x
// stages and deepen the weight ring to 5-7 chunks), and those variants pass the numerics tests,
// but up to 256 rows only ROWS = 256 is launched (the 128- and 64-row tiles serve the opt-in
// "narrow" row-block policy above 256 rows, see row_block): measured at 1 / 64 / 128 rows a decode
// step takes 3.73 / 4.30 / 5.41 ms with them against 3.58 / 4.31 / 5.41 ms with the register-ring kernels of
// gemm_skinny.hip, which already stream the large shapes at the HBM ceiling there (gate/up
// 5.3 TB/s at 64 rows) — a deeper ring has nothing left to hide.
// Weights are the packed-B image (ops/packing.py): a (n-tile, k-step) block is 64 lanes x 16 B in
// fragment order, so its LDS image is lane-linear and conflict-free; x rows are 128 B with the
// 16-B slot XOR-swizzled by (row >> 1) & 7, applied to the DMA source column and to the read.
//
// Replaces the nn.Linear calls inside HF LlamaDecoderLayer that the reference runs on CPU in
// fp32 (Worker1.py:128-137) and the lm_head (orchestration.py:141).
#include <atomic>
#include <cstdlib>
#include "common.h"
#include "gemm_common.h"
#include "launch.h"

namespace dlia {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int kR2Threads = 768;                   // 8 compute waves + 4 loader waves
constexpr int kR2XS = 3;                          // x stages
constexpr int kR2WSlot = 128 * 64 * 2;            // 16 KiB = 8 n-tiles x 2 k-steps x 1 KiB

template <int ROWS>
struct RowsCfg {
  static constexpr int MTW = ROWS / 32;           // m-tiles per compute wave (rows / 2 halves / 16)
  static constexpr int XP = ROWS / 32;            // x pieces (8 rows x 128 B) per loader and chunk
  static constexpr int XStage = ROWS * 128;       // bytes of one x stage
  static constexpr int WD = ROWS == 256 ? 3 : (ROWS == 128 ? 6 : (ROWS == 64 ? 7 : 8));   // weight ring slots
  static constexpr int LW = WD - 1;               // chunks the weight stream runs ahead
  static constexpr int Lds = kR2XS * XStage + WD * kR2WSlot;
  static_assert(2 * XP + 4 * LW <= 63, "vmcnt is a 6-bit counter");
};

// NP consecutive 1 KiB pieces, global -> LDS, under ONE setting of M0 (lane l of piece q lands at
// lds_addr + 1024 q + 16 l). The instruction's 13-bit signed immediate offset selects the piece; it
// advances the global AND the LDS address, so the caller passes every source pointer biased by
// minus r2_piece_off<NP>(q). A write to M0 waits until every earlier LDS-DMA instruction of the
// wave has read it (~180 cycles after issue): with M0 set, and restored, around every piece a wave
// issues one piece per ~180 cycles, with one setting per run one per ~50 (bench/l2_lds_bw.hip). M0
// is not restored: nothing else in these kernels uses it.
template <int NP> constexpr int r2_piece_off(int q) { return q * 1024 - (NP == 8 ? 4096 : 0); }
template <int NP>
__device__ __forceinline__ void r2_dma_run(const char* const (&g)[NP], unsigned lds_addr) {
  static_assert(NP == 1 || NP == 2 || NP == 4 || NP == 8, "pieces per run");
  if constexpr (NP == 8)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off offset:-4096\n\tglobal_load_lds_dwordx4 %2, off offset:-3072\n\t"
                 "global_load_lds_dwordx4 %3, off offset:-2048\n\tglobal_load_lds_dwordx4 %4, off offset:-1024\n\t"
                 "global_load_lds_dwordx4 %5, off\n\tglobal_load_lds_dwordx4 %6, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %7, off offset:2048\n\tglobal_load_lds_dwordx4 %8, off offset:3072"
                 :: "s"(lds_addr + 4096), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7])
                 : "memory", "m0");
  else if constexpr (NP == 4)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, off offset:2048\n\tglobal_load_lds_dwordx4 %4, off offset:3072"
                 :: "s"(lds_addr), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]) : "memory", "m0");
  else if constexpr (NP == 2)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024"
                 :: "s"(lds_addr), "v"(g[0]), "v"(g[1]) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_addr), "v"(g[0]) : "memory", "m0");
}
// four weight pieces (two n-tiles x two k-steps, 4 KiB in a row), non-temporal: the weight stream is
// read once and must not evict x from L2. g0 / g1: the n-tiles' sources, the second one passed 2048 low.
__device__ __forceinline__ void r2_dma_w4_nt(const void* g0, const void* g1_minus_2k, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off nt\n\tglobal_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
               "global_load_lds_dwordx4 %2, off offset:2048 nt\n\tglobal_load_lds_dwordx4 %2, off offset:3072 nt"
               :: "s"(lds_addr), "v"(g0), "v"(g1_minus_2k) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void r2_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ const void* r2_sel(bool c, const void* a, const void* b) {
  const long d = reinterpret_cast<const char*>(a) - reinterpret_cast<const char*>(b);
  return reinterpret_cast<const char*>(b) + (d & -static_cast<long>(c));
}

// Waves 0-7 compute; waves 8-11 only move data. A wave that issues DMA while the memory system is
// saturated blocks in the issue until the queue has room, and a blocked wave issues no MFMAs —
// with loader waves the matrix cores never see that back-pressure. Four loaders, because one
// wave can have at most 63 loads in flight (vmcnt is a 6-bit counter) and the CU needs ~112.
// F8: both operands are OCP e4m3 bytes (x rows of p.ldx BYTES with per-row scales p.xscale, W from
// pack_weight_a8w8 with per-column scales p.wscale). A 128-byte x row then holds 128 K-elements and
// the two 1 KiB weight blocks of a (n-tile, chunk) are the two halves of one 32-byte-per-lane
// fragment, so every byte count, LDS address and wait count below is the same as for bf16; only the
// chunk covers twice the K and is consumed by v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales,
// lane l = (r, q) holds k = 32 q .. 32 q + 31 of row r: 16-B slots 2q and 2q + 1 of its x row).
//
// DIAG: diagnostic build (DLIA_GEMM_ROWS_DIAG=1, never used for serving). Every wave accumulates
// s_memtime deltas of the sections of a chunk and lane 0 stores them to p.partial, which then is a
// debug buffer [workgroup][12 waves][8] of int64 that no other code reads: compute waves
// {barrier wait, first fragment reads, MFMA section, chunks, shader cycles and 100 MHz real-time
// ticks of the whole chunk loop (their ratio is the in-kernel clock)}, loaders {vmcnt wait, barrier
// wait, DMA issue, chunks}. The stamps cost ~10 % themselves; ratios are what they are for.
//
// What the stamps say (256 x 128 tile, gate/up shape, 256 and 4096 rows alike): a chunk takes
// ~1400 shader cycles for 1024 cycles of MFMA work per SIMD. Compute waves: ~190 at the barrier,
// ~240 waiting for their first fragment reads after it, ~710 in the MFMA section. Loaders: ~130
// waiting for data (it has landed), ~400 at the barrier, ~850 issuing their 12 DMA pieces
// (~70 cycles a piece). A schedule that lets compute waves prefetch the next chunk before the
// barrier — loaders guaranteeing chunk c+1 instead of c at barrier c — was tried and ran at 0.88
// instead of 1.15 PFLOP/s at 4096 rows: it costs one chunk of prefetch distance, and a piece takes
// longer than one chunk to land; LDS has no room for a fourth 32 KiB x stage. What the bf16 kernel
// does instead is hold back the last MFMA group of a chunk and execute it right after the next
// barrier, under the new chunk's first reads (HOLD below): 1.13-1.15 -> 1.17 PFLOP/s at 4096 rows —
// the SIMD partner wave was already covering most of that gap.
template <int ROWS, bool F8, bool DIAG = false>
__global__ __launch_bounds__(kR2Threads, 1) void gemm_rows_kernel(GemmParams p) {
  using C = RowsCfg<ROWS>;
  constexpr int XP = C::XP, MTW = C::MTW, LW = C::LW, WD = C::WD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  // More rows than ROWS: the grid also walks ROWS-row blocks. Workgroup ids go round-robin over the
  // 8 XCDs; an XCD runs 32 workgroups at a time, in step along K, and what they share they fetch
  // from beyond its 4 MiB L2 only once. Those 32 are made a patch of (32 / group_n) row blocks x
  // group_n column blocks (grouped order: within a group of group_n column blocks, ids walk the
  // columns first, then the rows): with 16 row blocks, group_n = 1 would stream all of x past L2
  // for every column block, group_n = 4 shares each x line four ways and each weight line eight.
  const int mblocks = (p.M + ROWS - 1) / ROWS;
  const int within = blockIdx.x >> 3;                       // index among this XCD's workgroups
  const int ncb = gridDim.x / (8 * mblocks);                // column blocks per XCD (grid padded to 8)
  const int grp = within / (p.group_n * mblocks);
  const int width = min(p.group_n, ncb - grp * p.group_n);
  const int in_grp = within - grp * p.group_n * mblocks;
  const int cb = grp * p.group_n + in_grp % width;
  const int nblk = cb * 8 + (blockIdx.x & 7), mrow0 = (in_grp / width) * ROWS;
  const int ks = blockIdx.y;
  if (nblk * 8 >= p.ntiles) return;                         // grid padded to a multiple of 8 column blocks
  const int chunks_total = p.K >> (F8 ? 7 : 6);             // chunk = 128 bytes of an x row
  const int KT = chunks_total * 2;                          // 1 KiB weight blocks per n-tile
  const int cper = (chunks_total + p.splitk - 1) / p.splitk;
  const int c0 = ks * cper;
  const int n = min(chunks_total, c0 + cper) - c0;          // chunks of this split
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  if (wave >= 8) {
    // ---------------- loader L: x rows [ROWS/4 L, +ROWS/4) and the weight blocks of n-quad L ----------------
    if (n <= 0) return;
    const int L = wave - 8;
    const void* const dummy = p.x;                          // 4 KiB of valid, L2-hot bytes
    // one DMA instruction fills 1 KiB: 8 x rows of 128 B (rows past M re-read the last row, never
    // stored to the output), or one (n-tile, k-step) weight block
    const char* xsrc[XP];
    const long ldxb = F8 ? p.ldx : p.ldx * 2;               // row stride in bytes
#pragma unroll
    for (int q = 0; q < XP; ++q) {
      const int row = 8 * (L * XP + q) + (lane >> 3);
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      xsrc[q] = reinterpret_cast<const char*>(p.x) + (long)min(mrow0 + row, p.M - 1) * ldxb + slot * 16 + (long)c0 * 128
                - r2_piece_off<XP>(q);
    }
    const u32x4* wsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wsrc[j] = p.wp + ((long)((nblk * 4 + L) * 2 + j) * KT + (long)c0 * 2) * 64 + lane;
    auto xdma = [&](int stage, int c) {
      const char* g[XP];
#pragma unroll
      for (int q = 0; q < XP; ++q) g[q] = xsrc[q] + (long)c * 128;
      r2_dma_run<XP>(g, lds0 + stage * C::XStage + L * XP * 1024);
    };
    auto wdma = [&](int slot, int c) {          // past the end of the split: 4 KiB of x instead (valid, L2-hot, never read back)
      r2_dma_w4_nt(r2_sel(c < n, wsrc[0] + (long)c * 128, dummy),
                   r2_sel(c < n, reinterpret_cast<const char*>(wsrc[1] + (long)c * 128) - 2048, dummy),
                   lds0 + kR2XS * C::XStage + slot * kR2WSlot + L * 4096);
    };
    // Per loader a chunk is XP x pieces and 4 weight pieces. Steady state: chunk c issues X(c+2)
    // then W(c+LW); the prologue issues W(0) .. W(LW-3) and then plays iterations -2 and -1.
    //  * LW >= 3: the loads younger than X(c) at the top of chunk c are W(c-2+LW) X(c+1) W(c-1+LW),
    //    so vmcnt(XP + 8) means this loader's pieces of X(c) and of the older W(c) have landed.
    //  * LW == 2: W(c) is issued right after X(c); younger than it are X(c+1) W(c+1) -> vmcnt(XP + 4).
    // The barrier extends "landed" to all loaders, and tells them every compute wave is done
    // reading chunk c-1, whose x stage and weight slot the new loads overwrite. Loads past the end
    // of the split re-read the last x chunk / a dummy line into buffers nobody reads again, so
    // the count holds to the last chunk.
#pragma unroll
    for (int i = 0; i + 2 < LW; ++i) wdma(i, i);
    xdma(0, 0); wdma(LW - 2, LW - 2);
    xdma(1, min(1, n - 1)); wdma(LW - 1, LW - 1);
    int xs = 0, wsl = 0;
    long d_vm = 0, d_bar = 0, d_issue = 0;
    for (int c = 0; c < n; ++c) {
      long t0 = 0, t1 = 0, t2 = 0;
      if constexpr (DIAG) t0 = __builtin_amdgcn_s_memtime();
      r2_wait_vm<LW >= 3 ? XP + 8 : XP + 4>();
      if constexpr (DIAG) t1 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      if constexpr (DIAG) t2 = __builtin_amdgcn_s_memtime();
      xdma(xs >= 1 ? xs - 1 : 2, min(c + 2, n - 1));          // stage (c + 2) % 3
      wdma(wsl >= 1 ? wsl - 1 : WD - 1, c + LW);              // slot (c + LW) % WD
      xs = xs == 2 ? 0 : xs + 1;
      wsl = wsl == WD - 1 ? 0 : wsl + 1;
      if constexpr (DIAG) {
        const long t3 = __builtin_amdgcn_s_memtime();
        d_vm += t1 - t0; d_bar += t2 - t1; d_issue += t3 - t2;
      }
    }
    r2_wait_vm<0>();                                        // no DMA may land after the workgroup is gone
    if constexpr (DIAG) {
      if (lane == 0) {
        long* d = reinterpret_cast<long*>(p.partial) + ((long)blockIdx.x * 12 + wave) * 8;
        d[0] = d_vm; d[1] = d_bar; d[2] = d_issue; d[3] = n;
      }
    }
    return;
  }

  // ---------------- compute wave (mh, nq): rows [ROWS/2 mh, +ROWS/2) x n-tiles [2 nq, +2) ----------------
  const int mh = wave >> 2, nq = wave & 3;
  const int tile0 = (nblk * 4 + nq) * 2;                    // (ntiles is a multiple of 8 by packing)
  f32x4 acc[MTW][2];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    // the x fragments are walked in groups of GS m-tiles, the next group's LDS reads in flight
    // while the MFMAs of the current one issue, across the two k-steps of the chunk
    constexpr int GS = MTW < 4 ? MTW : 4, GPK = MTW / GS, G = 2 * GPK;
    int xs = 0, wsl = 0;
    const int xoff = (mh * (ROWS / 2) + l15) * 128;
    const int sw = (l15 >> 1) & 7;
    long d_bar = 0, d_rd = 0, d_mfma = 0, loop_c0 = 0, loop_r0 = 0;
    if constexpr (DIAG) { loop_c0 = __builtin_amdgcn_s_memtime(); loop_r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr bool HOLD = !F8 && !DIAG && G >= 2 && (G % 2) == 0;   // see the HOLD branch below
    u32x4 hwf[2][2], hxg[2][GS];                              // HOLD: fragments carried across chunks
    if constexpr (HOLD) {                                     // chunk 0 has nothing held: zero operands add zero
#pragma unroll
      for (int j = 0; j < 2; ++j) hwf[1][j] = u32x4{0, 0, 0, 0};
#pragma unroll
      for (int i4 = 0; i4 < GS; ++i4) hxg[1][i4] = u32x4{0, 0, 0, 0};
    }
    for (int c = 0; c < n; ++c) {
      long t0 = 0, t1 = 0, t2 = 0;
      if constexpr (DIAG) t0 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DIAG) t1 = __builtin_amdgcn_s_memtime();
      const char* xb = smem + xs * C::XStage + xoff;
      const char* wb = smem + kR2XS * C::XStage + wsl * kR2WSlot + nq * 4096 + lane * 16;
      xs = xs == 2 ? 0 : xs + 1;
      wsl = wsl == WD - 1 ? 0 : wsl + 1;
      auto xfrag = [&](int i, int kk) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(xb + i * 2048 + (((kk * 4 + quad) ^ sw) << 4));
      };
      if constexpr (HOLD) {
        // Held-back last group: right after the barrier the matrix core would idle ~240 cycles until
        // the first fragment reads of the chunk return (stamps above). The last MFMA group of chunk
        // c-1 — k-step 1, m-tile group GPK-1, whose operands hw = wf[1] and hx = xg[1] were read from
        // chunk c-1's stage BEFORE this barrier and are complete (lgkmcnt(0) below) — is executed
        // here instead, under the new reads. Buffer roles repeat every chunk (G is even), so nothing
        // is copied: new wf[0] / xg[0] load first, the held MFMAs consume old wf[1] / xg[1], then
        // new wf[1] loads (needed from group GPK on) and group 1 prefetches into xg[1].
#pragma unroll
        for (int j = 0; j < 2; ++j) hwf[0][j] = *reinterpret_cast<const u32x4*>(wb + (j * 2) * 1024);
#pragma unroll
        for (int i4 = 0; i4 < GS; ++i4) hxg[0][i4] = xfrag(i4, 0);
        __builtin_amdgcn_sched_barrier(0);                    // the reads go out BEFORE the held MFMAs, not after
#pragma unroll
        for (int i4 = 0; i4 < GS; ++i4)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[(GPK - 1) * GS + i4][j] = mfma16(hwf[1][j], hxg[1][i4], acc[(GPK - 1) * GS + i4][j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) hwf[1][j] = *reinterpret_cast<const u32x4*>(wb + (j * 2 + 1) * 1024);
#pragma unroll
        for (int g = 0; g + 1 < G; ++g) {
          const int kk = g / GPK, ig = g % GPK;
#pragma unroll
          for (int i4 = 0; i4 < GS; ++i4) hxg[(g + 1) & 1][i4] = xfrag(((g + 1) % GPK) * GS + i4, (g + 1) / GPK);
#pragma unroll
          for (int i4 = 0; i4 < GS; ++i4)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[ig * GS + i4][j] = mfma16(hwf[kk][j], hxg[g & 1][i4], acc[ig * GS + i4][j]);
          __builtin_amdgcn_sched_barrier(0);
        }
        // group G-1 stays in hwf[1] / hxg[1]; its reads must be complete before the next barrier lets
        // the loaders overwrite this chunk's stage
        // (the builtin, not inline asm: hipcc's own wait insertion then knows the held fragments are
        // complete and does not drain the NEW reads before the held MFMAs of the next chunk;
        // 0xC07F = lgkmcnt(0) with vmcnt / expcnt left at "no wait")
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (F8) {
        // one MX MFMA per (m-tile, n-tile) and chunk; fragments are 32 bytes per lane
        i32x8 wf8[2], xq[2][GS];
        auto xfrag8 = [&](int i) -> i32x8 {
          const u32x4 lo = *reinterpret_cast<const u32x4*>(xb + i * 2048 + (((2 * quad) ^ sw) << 4));
          const u32x4 hi = *reinterpret_cast<const u32x4*>(xb + i * 2048 + (((2 * quad + 1) ^ sw) << 4));
          return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const u32x4 lo = *reinterpret_cast<const u32x4*>(wb + (j * 2) * 1024);
          const u32x4 hi = *reinterpret_cast<const u32x4*>(wb + (j * 2 + 1) * 1024);
          wf8[j] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int i4 = 0; i4 < GS; ++i4) xq[0][i4] = xfrag8(i4);
#pragma unroll
        for (int g = 0; g < GPK; ++g) {
          if (g + 1 < GPK) {
#pragma unroll
            for (int i4 = 0; i4 < GS; ++i4) xq[(g + 1) & 1][i4] = xfrag8((g + 1) * GS + i4);
          }
#pragma unroll
          for (int i4 = 0; i4 < GS; ++i4)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[g * GS + i4][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                  wf8[j], xq[g & 1][i4], acc[g * GS + i4][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      u32x4 wf[2][2], xg[2][GS];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[kk][j] = *reinterpret_cast<const u32x4*>(wb + (j * 2 + kk) * 1024);
#pragma unroll
      for (int i4 = 0; i4 < GS; ++i4) xg[0][i4] = xfrag(i4, 0);
      if constexpr (DIAG) {                                   // the first fragments have arrived
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        t2 = __builtin_amdgcn_s_memtime();
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {                           // (k-step, group of m-tiles)
        const int kk = g / GPK, ig = g % GPK;
        if (g + 1 < G) {
#pragma unroll
          for (int i4 = 0; i4 < GS; ++i4) xg[(g + 1) & 1][i4] = xfrag(((g + 1) % GPK) * GS + i4, (g + 1) / GPK);
        }
#pragma unroll
        for (int i4 = 0; i4 < GS; ++i4)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[ig * GS + i4][j] = mfma16(wf[kk][j], xg[g & 1][i4], acc[ig * GS + i4][j]);
        __builtin_amdgcn_sched_barrier(0);                    // keep the next group's reads ahead of these MFMAs
      }
      if constexpr (DIAG) {
        const long t3 = __builtin_amdgcn_s_memtime();
        d_bar += t1 - t0; d_rd += t2 - t1; d_mfma += t3 - t2;
      }
      }
    }
    if constexpr (HOLD) {                                     // the last chunk's held group
#pragma unroll
      for (int i4 = 0; i4 < GS; ++i4)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[(GPK - 1) * GS + i4][j] = mfma16(hwf[1][j], hxg[1][i4], acc[(GPK - 1) * GS + i4][j]);
    }
    if constexpr (DIAG) {
      if (lane == 0) {
        long* d = reinterpret_cast<long*>(p.partial) + ((long)blockIdx.x * 12 + wave) * 8;
        d[0] = d_bar; d[1] = d_rd; d[2] = d_mfma; d[3] = n;
        d[4] = __builtin_amdgcn_s_memtime() - loop_c0; d[5] = __builtin_amdgcn_s_memrealtime() - loop_r0;
      }
    }
  }
  if constexpr (F8) {
    // dequantise: per output column (weight row) and per row (token)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(p.wscale + (tile0 + j) * 16 + quad * 4);
#pragma unroll
      for (int i = 0; i < MTW; ++i) acc[i][j] *= sc;
    }
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const int m = mrow0 + mh * (ROWS / 2) + i * 16 + l15;
      const float xs = p.xscale[m < p.M ? m : p.M - 1];
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] *= xs;
    }
  }
  write_acc_tiles<MTW, 2>(p, acc, tile0, l15, quad, ks, mrow0 + mh * (ROWS / 2));
  // As the lm_head (plain fp32 output, no split): the wave's two n-tiles are one 32-column chunk of the logits row;
  // its maximum over the columns below n_valid goes to chunk_max[m][chunk] for the guided sampler.
  if constexpr (!F8) {
    if (p.chunk_max != nullptr && p.splitk == 1) {
      const int chunk = tile0 >> 1;
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        float mx = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((tile0 + j) * 16 + quad * 4 + r < p.n_valid) mx = fmaxf(mx, acc[i][j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const int m = mrow0 + mh * (ROWS / 2) + i * 16 + l15;
        if (quad == 0 && m < p.M) p.chunk_max[(long)m * p.ldc + chunk] = mx;
      }
    }
  }
}

// Column blocks per workgroup-order group (see the kernel). Measured: 4 is worth +2.7 % at 4096 rows
// (16 row blocks), nothing at 2048, and costs 1.5 % on the serving operating point (1024 rows, two
// micro-batches on two streams: 55.5k vs 56.4k tok/s, same device, interleaved runs) — so 4 from 16
// row blocks up, 1 (column block by column block) below. DLIA_GEMM_ROWS_GROUP overrides (A/B runs).
static int order_group(int mblocks) {
  static const int forced = [] { const char* e = getenv("DLIA_GEMM_ROWS_GROUP"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 32 ? v : 0; }();
  return forced ? forced : (mblocks >= 16 ? 4 : 1);
}

template <int ROWS, bool F8>
static void launch_rows(const GemmParams& p, hipStream_t stream) {
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows_kernel<ROWS, F8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, RowsCfg<ROWS>::Lds);
    return true;
  }();
  (void)once;
  const int nblocks8 = (p.ntiles / 8 + 7) / 8 * 8, mblocks = (p.M + ROWS - 1) / ROWS;
  GemmParams q = p;
  q.group_n = order_group(mblocks);
  hipLaunchKernelGGL((gemm_rows_kernel<ROWS, F8>), dim3(nblocks8 * mblocks, p.splitk), dim3(kR2Threads), RowsCfg<ROWS>::Lds,
                     stream, q);
}

// Row-block size for more than 256 rows. 256-row blocks do the most work per DMA byte, but a shape
// with few column blocks then launches few workgroups (Llama-3-8B O / down projections at 1024
// rows: 32 column blocks x 4 row blocks = 128 for 256 CUs). Smaller blocks multiply the workgroups;
// the "narrow" policy takes the largest block that still gives ~200 workgroups. Measured: a GEMM
// running ALONE gains (layer total at 512 / 1024 rows 398 -> 284 / 513 -> 437 us), but the serving
// path runs two or three micro-batches on separate streams, and there the narrow launch that left
// half the CUs to the other stream's kernels was the better one: bench.py 3 x 512 rows 53.3k ->
// 44.0k tok/s, 2 x 1024 rows 56k -> 54k with the policy on. So it is OFF unless selected
// (DLIA_GEMM_ROWS_NARROW=1, or dlia_gemm_rows_set_narrow / the large_gemm_kernel op at run time).
static std::atomic<int>& narrow_policy() {
  static std::atomic<int> v{[] { const char* e = getenv("DLIA_GEMM_ROWS_NARROW"); return (e && atoi(e) != 0) ? 1 : 0; }()};
  return v;
}
static int row_block(const GemmParams& p) {
  if (!narrow_policy().load() || p.M <= 256) return 256;
  const int nb = p.ntiles / 8;
  if (nb * ((p.M + 255) / 256) > 160) return 256;
  if (nb * ((p.M + 127) / 128) > 160) return 128;
  return 64;
}

}  // namespace dlia

using namespace dlia;

// Called by dlia_gemm_skinny for bf16 weights and M > 128; `p` is fully set up (split-K only up to
// 256 rows).
extern "C" void dlia_gemm_rows(const dlia::GemmParams* p, hipStream_t stream) {
  static const bool diag = [] { const char* e = getenv("DLIA_GEMM_ROWS_DIAG"); return e && atoi(e) != 0; }();
  if (diag && p->partial && p->splitk == 1 && !p->partial_only) {       // p->partial = debug buffer, see DIAG
    static const bool once = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows_kernel<256, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, RowsCfg<256>::Lds);
      return true;
    }();
    (void)once;
    const int nblocks8 = (p->ntiles / 8 + 7) / 8 * 8, mblocks = (p->M + 255) / 256;
    GemmParams q = *p;
    q.group_n = order_group(mblocks);
    hipLaunchKernelGGL((gemm_rows_kernel<256, false, true>), dim3(nblocks8 * mblocks, 1), dim3(kR2Threads),
                       RowsCfg<256>::Lds, stream, q);
    return;
  }
  switch (row_block(*p)) {
    case 64:  launch_rows<64, false>(*p, stream); break;
    case 128: launch_rows<128, false>(*p, stream); break;
    default:  launch_rows<256, false>(*p, stream);
  }
}

// The same for e4m3 x e4m3 (called by dlia_gemm_a8w8 for M > 128): p->x = e4m3 rows (ldx in
// bytes), p->xscale per row, p->wp from pack_weight_a8w8, p->wscale per packed row.
extern "C" void dlia_gemm_rows_f8(const dlia::GemmParams* p, hipStream_t stream) {
  switch (row_block(*p)) {
    case 64:  launch_rows<64, true>(*p, stream); break;
    case 128: launch_rows<128, true>(*p, stream); break;
    default:  launch_rows<256, true>(*p, stream);
  }
}

// 1: pick 128- / 64-row blocks for shapes with few column blocks (see row_block); returns the old value.
extern "C" int dlia_gemm_rows_set_narrow(int on) { return narrow_policy().exchange(on ? 1 : 0); }
