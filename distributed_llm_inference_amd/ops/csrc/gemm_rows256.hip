// Weight-streaming bf16 GEMM for 129..256 activation rows: out[M,N] = epi(x[M,K] @ W[N,K]^T).
//
// At this row count the launch is bound by neither HBM nor the matrix cores alone but by keeping
// two streams in flight per CU at once: every workgroup re-stages all of x from L2 (470 MB for the
// Llama-3-8B gate/up shape) while it streams its own weight columns from HBM (235 MB). Measured on
// MI355X with the compute removed, x alone moves at the L2 ceiling (19 TB/s), the weights alone at
// the HBM ceiling (5.5 TB/s), both together in ~55 us; the kernel's job is to add as little as
// possible to that. It therefore
//  * moves BOTH operands global -> LDS by DMA (no register ring): three 32 KiB stages of x
//    (prefetched two chunks ahead) and a four-deep ring of 16 KiB weight slots (three chunks =
//    48 KiB per CU ahead) — all 160 KiB of LDS — with one hand-counted vmcnt + one barrier per
//    64-wide K chunk; the weight DMA is non-temporal so the stream cannot evict x from L2;
//  * splits the workgroup into 8 compute waves (two per SIMD) on a 256 x 128 tile — wave (mh, nq)
//    owns rows [128 mh, +128) x n-tiles [2 nq, +2), 64 accumulator registers, the two row halves
//    sharing each weight fragment through the LDS ring — and 4 loader waves that do nothing but
//    issue the DMA: with the loads issued by the compute waves themselves the same kernel took
//    72 us instead of 59 (a wave blocked in a full memory queue issues no MFMAs).
// Weights are the packed-B image (ops/packing.py): a (n-tile, k-step) block is 64 lanes x 16 B in
// fragment order, so its LDS image is lane-linear and conflict-free; x rows are 128 B with the
// 16-B slot XOR-swizzled by (row >> 1) & 7, applied to the DMA source column and to the read.
//
// Replaces the nn.Linear calls inside HF LlamaDecoderLayer that the reference runs on CPU in
// fp32 (Worker1.py:128-137) and the lm_head (orchestration.py:141) for large decode batches.
#include "common.h"
#include "gemm_common.h"
#include "launch.h"

namespace dlia {

constexpr int kR2Threads = 768;                   // 8 compute waves + 4 loader waves
constexpr int kR2XS = 3;                          // x stages
constexpr int kR2WD = 3;                          // weight ring depth (chunks)
constexpr int kR2XStage = 256 * 64 * 2;           // 32 KiB
constexpr int kR2WSlot = 128 * 64 * 2;            // 16 KiB = 8 n-tiles x 2 k-steps x 1 KiB
constexpr int kR2Lds = kR2XS * kR2XStage + kR2WD * kR2WSlot;

// 16 B per lane, global -> LDS: lane l lands at lds_addr + 16 l. M0 carries the wave-uniform LDS
// address and is restored, since the compiler does not expect inline asm to change it.
__device__ __forceinline__ void r2_dma16(const void* gptr, unsigned lds_addr) {
  unsigned saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
               "s_nop 0\n\ts_mov_b32 m0, %0"
               : "=&s"(saved) : "s"(lds_addr), "v"(gptr) : "memory");
}
// the same, non-temporal: the weight stream is read once and must not evict x from L2
__device__ __forceinline__ void r2_dma16_nt(const void* gptr, unsigned lds_addr) {
  unsigned saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\t"
               "s_nop 0\n\ts_mov_b32 m0, %0"
               : "=&s"(saved) : "s"(lds_addr), "v"(gptr) : "memory");
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
__global__ __launch_bounds__(kR2Threads, 1) void gemm_rows256_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  // More than 256 rows: the grid also walks 256-row blocks. Workgroup ids go round-robin over the
  // 8 XCDs, so the row blocks of one column block get ids that differ by multiples of 8: they run
  // on the same XCD at about the same time and share each weight byte through its L2.
  const int mblocks = (p.M + 255) >> 8;
  const int within = blockIdx.x >> 3;
  const int nblk = (within / mblocks) * 8 + (blockIdx.x & 7), mrow0 = (within % mblocks) * 256;
  const int ks = blockIdx.y;
  if (nblk * 8 >= p.ntiles) return;                         // grid padded to a multiple of 8 column blocks
  const int KT = p.K >> 5;
  const int chunks_total = p.K >> 6;
  const int cper = (chunks_total + p.splitk - 1) / p.splitk;
  const int c0 = ks * cper;
  const int n = min(chunks_total, c0 + cper) - c0;          // chunks of this split
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  if (wave >= 8) {
    // ---------------- loader L: x rows [64 L, +64) and the weight blocks of n-quad L ----------------
    if (n <= 0) return;
    const int L = wave - 8;
    const void* const dummy = p.x;                          // 16 valid, L2-hot bytes
    // one DMA instruction fills 1 KiB: 8 x rows of 128 B (rows past M re-read the last row, never
    // stored to the output), or one (n-tile, k-step) weight block
    const bf16* xsrc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = 8 * (L * 8 + q) + (lane >> 3);
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      xsrc[q] = p.x + (long)min(mrow0 + row, p.M - 1) * p.ldx + slot * 8 + (long)c0 * 64;
    }
    const u32x4* wsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wsrc[j] = p.wp + ((long)((nblk * 4 + L) * 2 + j) * KT + (long)c0 * 2) * 64 + lane;
    auto xdma = [&](int stage, int c) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        r2_dma16(xsrc[q] + (long)c * 64, lds0 + stage * kR2XStage + (L * 8 + q) * 1024);
    };
    auto wdma = [&](int slot, int c) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          r2_dma16_nt(r2_sel(c < n, wsrc[j] + (long)(c * 2 + kk) * 64, dummy),
                      lds0 + kR2XS * kR2XStage + slot * kR2WSlot + ((L * 2 + j) * 2 + kk) * 1024);
    };
    // kR2WD == 4 — issue order: W(0) | X(0) W(1) | X(1) W(2) | then chunk c issues X(c+2) W(c+3).
    // At the top of chunk c the loads younger than X(c) are W(c+1) X(c+1) W(c+2) = 4 + 8 + 4, so
    // vmcnt(16) means this loader's pieces of X(c) and of the older W(c) have landed.
    // kR2WD == 3 — X(0) W(0) | X(1) W(1) | then X(c+2) W(c+2); younger than W(c) are X(c+1)
    // W(c+1) = 8 + 4 -> vmcnt(12).
    // The barrier extends "landed" to all loaders, and tells them every compute wave is done
    // reading chunk c-1, whose x stage and weight slot the new loads overwrite. Loads past the end
    // of the split re-read the last x chunk / a dummy line into buffers nobody reads again, so
    // the count holds to the last chunk.
    constexpr int LW = kR2WD - 1;                           // chunks the weight stream runs ahead
    if constexpr (LW == 3) {
      wdma(0, 0);
      xdma(0, 0); wdma(1, 1);
      xdma(1, min(1, n - 1)); wdma(2, 2);
    } else {
      xdma(0, 0); wdma(0, 0);
      xdma(1, min(1, n - 1)); wdma(1, 1);
    }
    int xs = 0, wsl = 0;
    for (int c = 0; c < n; ++c) {
      r2_wait_vm<LW == 3 ? 16 : 12>();
      __builtin_amdgcn_s_barrier();
      xdma(xs >= 1 ? xs - 1 : 2, min(c + 2, n - 1));          // stage (c + 2) % 3
      wdma(wsl >= 1 ? wsl - 1 : kR2WD - 1, c + LW);           // slot (c + LW) % kR2WD
      xs = xs == 2 ? 0 : xs + 1;
      wsl = wsl == kR2WD - 1 ? 0 : wsl + 1;
    }
    r2_wait_vm<0>();                                        // no DMA may land after the workgroup is gone
    return;
  }

  // ---------------- compute wave (mh, nq): rows [128 mh, +128) x n-tiles [2 nq, +2) ----------------
  const int mh = wave >> 2, nq = wave & 3;
  const int tile0 = (nblk * 4 + nq) * 2;                    // (ntiles is a multiple of 8 by packing)
  f32x4 acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    int xs = 0, wsl = 0;
    const int xoff = (mh * 128 + l15) * 128;
    const int sw = (l15 >> 1) & 7;
    for (int c = 0; c < n; ++c) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* xb = smem + xs * kR2XStage + xoff;
      const char* wb = smem + kR2XS * kR2XStage + wsl * kR2WSlot + nq * 4096 + lane * 16;
      xs = xs == 2 ? 0 : xs + 1;
      wsl = wsl == kR2WD - 1 ? 0 : wsl + 1;
      auto xfrag = [&](int i, int kk) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(xb + i * 2048 + (((kk * 4 + quad) ^ sw) << 4));
      };
      u32x4 wf[2][2], xg[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[kk][j] = *reinterpret_cast<const u32x4*>(wb + (j * 2 + kk) * 1024);
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) xg[0][i4] = xfrag(i4, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {                           // (k-step, half of the m-tiles)
        const int kk = g >> 1, ig = g & 1;
        if (g + 1 < 4) {
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) xg[(g + 1) & 1][i4] = xfrag(((g + 1) & 1) * 4 + i4, (g + 1) >> 1);
        }
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[ig * 4 + i4][j] = mfma16(wf[kk][j], xg[g & 1][i4], acc[ig * 4 + i4][j]);
        __builtin_amdgcn_sched_barrier(0);                    // keep the next group's reads ahead of these MFMAs
      }
    }
  }
  write_acc_tiles<8, 2>(p, acc, tile0, l15, quad, ks, mrow0 + mh * 128);
}

}  // namespace dlia

using namespace dlia;

// Called by dlia_gemm_skinny for bf16 weights and M > 128 (any number of 256-row blocks); `p` is fully set up.
extern "C" void dlia_gemm_rows256(const dlia::GemmParams* p, hipStream_t stream) {
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows256_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kR2Lds);
    return true;
  }();
  (void)once;
  const int nblocks8 = (p->ntiles / 8 + 7) / 8 * 8, mblocks = (p->M + 255) / 256;
  hipLaunchKernelGGL(gemm_rows256_kernel, dim3(nblocks8 * mblocks, p->splitk), dim3(kR2Threads), kR2Lds, stream, *p);
}
