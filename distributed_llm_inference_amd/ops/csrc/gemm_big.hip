// bf16 GEMM for many rows (prefill chunks, large decode batches): out[M,N] = epi(x[M,K] @ W[N,K]^T)
// on 256 x 256 tiles, K walked in 32-wide steps.
//
// With hundreds of rows both operands come out of L2 / Infinity Cache. A 256 x 128 tile
// (gemm_rows.hip) moves 48 KiB per 64 K-elements; a 256 x 256 tile moves 64 KiB for twice the
// work, and a compute wave that holds all four weight fragments of a k-step issues 12 LDS fragment
// reads per 32 MFMAs instead of 20. Its accumulators are half of the CU's register file (8 compute
// waves x 128 registers), so everything else is kept small enough (163 VGPRs) for three waves per
// SIMD: the DMA is issued by 4 loader waves that share the SIMDs with the compute waves.
// Measured on MI355X (Llama-3-8B shapes): 1.18 PFLOP/s at 4096 rows, 0.96 at 2048 — the 256 x 128
// kernel does 1.15 / 1.13 and stays the default; counters show the matrix cores busy 47 % of the
// time in both, no LDS bank conflicts and < 5 % LDS waits, i.e. neither kernel is bound by LDS or
// by DMA bytes but by the wait + barrier structure of a k-step.
//  * LDS: five 32 KiB stages (16 KiB of x + 16 KiB of W per k-step), filled four k-steps ahead by
//    global -> LDS DMA, one hand-counted vmcnt + one barrier per k-step.
//  * Both LDS images are in MFMA-fragment order (block = 64 lanes x 16 B, lane l holds row l & 15,
//    k-elements 8 (l >> 4) .. +8), so every ds_read_b128 is lane-linear and conflict-free. W is
//    stored that way in memory (packed-B, ops/packing.py); for x the DMA source address of lane l
//    is simply row (l & 15), column 8 (l >> 4) of the block.
//  * Wave (mh, nq) of the 2 x 4 compute grid owns rows [128 mh, +128) x n-tiles [4 nq, +4).
//  * Workgroup ids go round-robin over the 8 XCDs; the row blocks of one column block get ids that
//    differ by multiples of 8, so they share each weight byte through one XCD's L2.
//
// Replaces the nn.Linear calls of HF LlamaDecoderLayer that the reference runs on CPU in fp32 over
// the whole sequence (Worker1.py:128-137) and the lm_head (orchestration.py:141).
#include "common.h"
#include "gemm_common.h"
#include "launch.h"

namespace dlia {

constexpr int kBigThreads = 768;                  // 8 compute waves + 4 loader waves
constexpr int kBigStages = 5;
constexpr int kBigLead = kBigStages - 1;          // k-steps the loaders run ahead
constexpr int kBigStage = 32 * 1024;              // 16 x-blocks + 16 W-blocks of 1 KiB
constexpr int kBigLds = kBigStages * kBigStage;

__device__ __forceinline__ void big_dma16(const void* gptr, unsigned lds_addr) {
  unsigned saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
               "s_nop 0\n\ts_mov_b32 m0, %0"
               : "=&s"(saved) : "s"(lds_addr), "v"(gptr) : "memory");
}
template <int N>
__device__ __forceinline__ void big_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(kBigThreads, 1) void gemm_big_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const int mblocks = (p.M + 255) >> 8;
  const int within = blockIdx.x >> 3;
  const int nblk = (within / mblocks) * 8 + (blockIdx.x & 7), mrow0 = (within % mblocks) * 256;
  if (nblk * 16 >= p.ntiles) return;                        // grid padded to a multiple of 8 column blocks
  const int KT = p.K >> 5;                                  // k-steps in K
  const int n = KT;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  if (wave >= 8) {
    // ---------------- loader L: x row blocks 4L..4L+3 and W n-tiles 4L..4L+3 of the tile ----------------
    const int L = wave - 8;
    const bf16* xsrc[4];
    const u32x4* wsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = mrow0 + (L * 4 + q) * 16 + l15;       // rows past M re-read the last row (never stored)
      xsrc[q] = p.x + (long)min(row, p.M - 1) * p.ldx + quad * 8;
      const int tile = min(nblk * 16 + L * 4 + q, p.ntiles - 1);   // (ntiles is a multiple of 8, not of 16)
      wsrc[q] = p.wp + (long)tile * KT * 64 + lane;
    }
    auto fill = [&](int stage, int c) {
      const unsigned base = lds0 + stage * kBigStage + L * 4096;
#pragma unroll
      for (int q = 0; q < 4; ++q) big_dma16(xsrc[q] + (long)c * 32, base + q * 1024);
#pragma unroll
      for (int q = 0; q < 4; ++q) big_dma16(wsrc[q] + (long)c * 64, base + 16384 + q * 1024);
    };
    // k-step c is issued kBigLead steps ahead, 8 loads per step and loader: at the top of step c
    // the loads younger than step c's are those of steps c+1 .. c+kBigLead-1, so
    // vmcnt(8 (kBigLead-1)) means this loader's part of step c has landed; the barrier extends
    // that to all loaders and tells them the compute waves are done with step c-1, whose stage
    // the next fill overwrites. Steps past the end re-read the last one into a stage nobody
    // reads again, which keeps the count valid to the end.
#pragma unroll
    for (int c = 0; c < kBigLead; ++c) fill(c, min(c, n - 1));
    int st = 0;
    for (int c = 0; c < n; ++c) {
      big_wait_vm<8 * (kBigLead - 1)>();
      __builtin_amdgcn_s_barrier();
      fill(st == 0 ? kBigStages - 1 : st - 1, min(c + kBigLead, n - 1));   // stage (c + lead) % stages
      st = st == kBigStages - 1 ? 0 : st + 1;
    }
    big_wait_vm<0>();                                       // no DMA may land after the workgroup is gone
    return;
  }

  // ---------------- compute wave (mh, nq) ----------------
  const int mh = wave >> 2, nq = wave & 3;
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    int st = 0;
    const int xo = mh * 8192 + lane * 16, wo = 16384 + nq * 4096 + lane * 16;
    for (int c = 0; c < n; ++c) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* xb = smem + st * kBigStage + xo;
      const char* wb = smem + st * kBigStage + wo;
      st = st == kBigStages - 1 ? 0 : st + 1;
      // all four weight fragments of the k-step stay in registers and x streams past them in pairs
      // of m-tiles, the next pair in flight: 12 fragment reads per 32 MFMAs
      u32x4 wf[4], xg[2][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const u32x4*>(wb + j * 1024);
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) xg[0][i2] = *reinterpret_cast<const u32x4*>(xb + i2 * 1024);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g + 1 < 4) {
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
            xg[(g + 1) & 1][i2] = *reinterpret_cast<const u32x4*>(xb + ((g + 1) * 2 + i2) * 1024);
        }
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[g * 2 + i2][j] = mfma16(wf[j], xg[g & 1][i2], acc[g * 2 + i2][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  write_acc_tiles<8, 4>(p, acc, nblk * 16 + nq * 4, l15, quad, 0, mrow0 + mh * 128);
}

}  // namespace dlia

using namespace dlia;

// Same contract as dlia_gemm_tile (no split-K, fused epilogue), M > 256.
extern "C" void dlia_gemm_big(void* out, const void* x, const void* wp, const void* bias,
                              const void* residual, int M, int N, int K, long ldx, long ldo, long ldr,
                              int epi, int out_f32, hipStream_t stream) {
  if (M <= 0 || N <= 0) return;
  GemmParams p;
  p.x = (const bf16*)x; p.ldx = ldx; p.wp = (const u32x4*)wp; p.wscale = nullptr; p.xscale = nullptr; p.out = out; p.ldo = ldo;
  p.bias = (epi == 1 || epi == 3 || epi == 5) ? (const bf16*)bias : nullptr;
  p.residual = (epi == 2 || epi == 3) ? (const bf16*)residual : nullptr;
  p.ldr = ldr; p.partial = nullptr; p.M = M; p.N = N; p.K = K;
  p.ntiles = packed_ntiles(N);
  p.splitk = 1;
  p.act = epi == 4 ? 1 : (epi == 5 ? 2 : 0);
  p.out_f32 = out_f32;
  p.partial_only = 0;
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kBigLds);
    return true;
  }();
  (void)once;
  const int nblocks8 = ((p.ntiles + 15) / 16 + 7) / 8 * 8, mblocks = (M + 255) / 256;
  hipLaunchKernelGGL(gemm_big_kernel, dim3(nblocks8 * mblocks), dim3(kBigThreads), kBigLds, stream, p);
}
