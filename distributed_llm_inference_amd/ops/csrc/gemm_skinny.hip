// Weight-streaming GEMM for the decode regime: out[M,N] = epi(x[M,K] @ W[N,K]^T), M <= 256.
//
// Regime: every weight byte is read once per step and M is tiny, so the kernel is a
// HBM-streaming loop whose only job is to keep ~32+ KiB of loads in flight per CU.
//
//  * W is pre-packed at load time in MFMA-fragment order ("packed-B", see ops/packing.py):
//    block (n-tile nt, k-step kt) = 64 lanes x 16 B holding, for lane l,
//    W[nt*16 + (l&15)][kt*32 + (l>>4)*8 .. +8]. Blocks are stored [nt][kt], so a wave walking K
//    for one n-tile reads one contiguous stream, 1 KiB per wave-instruction, straight into the
//    VGPRs that feed v_mfma_f32_16x16x32_bf16 (no LDS round trip for the streamed operand),
//    with non-temporal loads so the stream does not evict the reusable activation rows.
//  * x (M x K, L2-resident, re-read by every workgroup) goes through LDS in full 256-B rows,
//    XOR-swizzled by (row & 15) on the 16-B slot so both the ds_write_b128 fill and the
//    ds_read_b128 fragment reads are bank-conflict-free.
//  * MFMA roles are swapped (A = W fragment, B = x fragment) so that each lane ends up with 4
//    consecutive output columns of one row: 8-byte bf16 / 16-byte fp32 stores.
//  * Small N x large K shapes do not have enough column blocks to fill 256 CUs; those are split
//    along K across workgroups, each writing an fp32 partial [S][M][N]. The partials are summed
//    by whichever kernel consumes the result next (the norm / rope kernels take partial inputs)
//    or by the reduce kernel below — at a launch boundary, where no fence protocol is needed.
//
//  * fp8 weights (W8): the same stream at half the bytes. A 16-B lane load holds OCP e4m3 values
//    for TWO k-steps (bytes 0-7: k-step 2t, bytes 8-15: k-step 2t+1, same (row, k) lane mapping);
//    they are widened in registers (v_cvt_pk_f32_fp8 + v_perm: every e4m3 value is exact in bf16)
//    and fed to the same bf16 MFMA, so activations keep bf16 precision. The per-output-channel
//    scale is applied to the fp32 accumulators before the epilogue.
//
// Replaces the nn.Linear calls inside HF LlamaDecoderLayer that the reference runs on CPU in
// fp32 over the whole sequence each step (Worker1.py:128-137) and the lm_head
// (orchestration.py:141).
#include <cstdlib>
#include "common.h"
#include "gemm_common.h"
#include "launch.h"

namespace dlia {

constexpr int kSkinnyBK = 128;       // K elements per LDS stage (4 MFMA k-steps)
constexpr int kSkinnyThreads = 256;  // 4 waves

template <int V> struct IntC { static constexpr int value = V; };

// The loads of the main loop are issued through inline asm and waited for with hand-counted
// s_waitcnt vmcnt(N): hipcc's own wait insertion loses track of loads that stay in flight across
// the loop back-edge and drains the queue every chunk, which collapses the prefetch depth. The
// counter is in-order, so "all but the N youngest are done" is exact as long as EVERY load in
// the loop goes through these helpers (the compiler then emits no vmcnt of its own in there).
//
// The compiler believes a destination is valid as soon as the load statement has executed, so
// the contract has two halves: (1) no consumer may be scheduled above its wait — the wait
// clobbers memory (orders the LDS stores) and is followed by a scheduling barrier (orders the
// MFMAs); (2) the compiler must not COPY a destination between load and wait. A register-tied
// wait operand provoked exactly such copies in the peeled tail, so the wait names no registers.
// ops/isa_lint.py re-checks (2) on the generated ISA at build time.
__device__ __forceinline__ void gload16(u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}
__device__ __forceinline__ void gload16_nt(u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p));
}
// `c ? a : b` on pointers without a branch: hipcc otherwise branches around the load that follows,
// and a load that sits on one arm of a diamond can no longer be counted by hand (nor by the lint).
__device__ __forceinline__ const void* sel_ptr(bool c, const void* a, const void* b) {
  const long d = reinterpret_cast<const char*>(a) - reinterpret_cast<const char*>(b);
  return reinterpret_cast<const char*>(b) + (d & -static_cast<long>(c));
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// BK = K elements per chunk (one LDS stage of x). 128 everywhere except the 256-row variant, whose
// two 64 KiB stages would allow only one workgroup — one wave per SIMD — per CU; with BK = 64 two
// workgroups share a CU and each covers the other's barrier, LDS and load stalls.
// OCC = waves per SIMD the register budget is sized for.
template <int MT, int NTW, bool W8, int BK, int OCC>
__global__ __launch_bounds__(kSkinnyThreads, OCC) void gemm_skinny_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages of MT*16 rows x BK bf16
  constexpr int KSTEPS = BK / 32;                // MFMA k-steps per chunk
  constexpr int ROWB = BK * 2;                   // bytes of one x row in an LDS stage
  constexpr int kStageBytes = MT * 16 * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int nblk = blockIdx.x, ks = blockIdx.y;
  constexpr int SL = W8 ? KSTEPS / 2 : KSTEPS;   // 16-B ring slots per chunk and n-tile
  constexpr int KPS = KSTEPS / SL;               // MFMA k-steps held by one slot
  const int KT = p.K / (32 * KPS);               // slots in K per n-tile
  const int chunks_total = p.K / BK;
  const int cper = (chunks_total + p.splitk - 1) / p.splitk;
  const int c0 = ks * cper;
  const int c1 = min(chunks_total, c0 + cper);
  const int n = c1 - c0;                         // LDS stages (chunks of BK K-elements) of this split
  // 16-B slot swizzle of an x row in LDS: 256-B rows all start at bank 0, so 16 consecutive rows
  // take 16 different slots; 128-B rows alternate between two bank halves, so row pairs do.
  auto swz = [](int row) { return BK == 128 ? (row & 15) : ((row >> 1) & 7); };
  const int tile0 = (nblk * 4 + wave) * NTW;     // first n-tile of this wave
  const bool wave_active = tile0 < p.ntiles;     // (ntiles is a multiple of 8 by packing)

  // x staging map: thread -> (row = tid/16 + 16 i, 16-B slot = tid%16), slot XOR-swizzled by row&15.
  // Row r = srow + 16 i has r & 15 == srow & 15, so the LDS destination of piece i is a constant
  // 4 KiB step from piece 0 (an immediate offset) and only one source pointer is kept; rows past
  // M re-read the last row (never stored to the output).
  const int srow = tid >> 4, sslot = tid & 15;
  const bf16* xbase = p.x + sslot * 8 + (long)c0 * BK;
  const int xdst0 = srow * 256 + ((sslot ^ (srow & 15)) << 4);
  auto xsrc = [&](int i, int c) -> const bf16* {
    const int r = srow + 16 * i;
    return xbase + (long)(r < p.M ? r : p.M - 1) * p.ldx + (long)c * BK;
  };
  // 256-row variants (MT == 16): the register file cannot also hold a chunk of x, so x goes
  // global -> LDS by DMA. One wave instruction fills 1 KiB = RPI rows; wave instruction
  // Q = wave*XL + q fills rows RPI*Q .. RPI*Q + RPI-1. The LDS image must be lane-linear, so the XOR
  // swizzle is applied to the SOURCE column instead.
  constexpr bool XDMA = MT >= 16;
  // At two waves per SIMD the peeled tail bodies (each a separately register-allocated copy of 64+
  // MFMAs) make hipcc shuffle accumulators through scratch at every join. That variant keeps ONE
  // body per chunk parity instead: loads for chunks past the end of the split are still issued,
  // against a dummy address, so the hand-counted waits stay valid to the last chunk.
  constexpr bool UNIFORM = OCC >= 2;
  const void* const dummy = p.x;                 // 16 valid, L2-hot bytes; every lane reads the same line
  static_assert(XDMA || BK == 128, "register staging is laid out for 256-B rows");
  constexpr int RPI = 1024 / ROWB;                       // rows per DMA instruction (4 or 8)
  constexpr int LPR = ROWB / 16;                         // lanes (16-B slots) per row (16 or 8)
  constexpr int XL = MT * 16 / RPI / 4;                  // DMA instructions per wave and chunk
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  auto xdma = [&](int stage, int c) {
    const void* src[XL];
#pragma unroll
    for (int q = 0; q < XL; ++q) {
      const int row = RPI * (wv * XL + q) + lane / LPR;
      src[q] = p.x + (long)(row < p.M ? row : p.M - 1) * p.ldx + (((lane % LPR) ^ swz(row)) << 3) + (long)(c0 + c) * BK;
    }
    lds_dma_pieces<XL>(src, lds0 + stage * kStageBytes + wv * XL * 1024);
  };
  const u32x4* wbase[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
    wbase[j] = p.wp + ((long)(wave_active ? tile0 + j : 0) * KT + (long)c0 * SL) * 64 + lane;

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (n > 0) {
    // Weight ring: 2 chunks (8 k-steps) deep. Slot (par*4 + kk) holds k-step kk of the chunk with
    // parity par and is refilled with the chunk two ahead right after its last use, so up to
    // 16 KiB of weights per wave stay in flight — also across the per-chunk barrier, which only
    // orders the two LDS stages of x.
    // Issue order (per chunk c, after the prologue): x(c+2) [MT loads], then per slot s the
    // refill of slot s [NTW loads]. Hence, in steady state (SL slots per chunk):
    //   x(c+1) has SL*NTW younger loads when chunk c starts              -> vmcnt(SL*NTW)
    //   slot s of chunk c has (2*SL-1)*NTW + 2*MT younger loads at its use -> vmcnt of that
    constexpr int XLOADS = XDMA ? XL : MT;           // x loads per wave (thread) and chunk
    constexpr int kWaitX = SL * NTW, kWaitW = (2 * SL - 1) * NTW + 2 * XLOADS;
    static_assert(kWaitW <= 63, "vmcnt is a 6-bit counter");
    u32x4 w[2 * SL][NTW], xr[MT];
    if constexpr (!XDMA) {
      // prologue, in the same order as the steady state: x(0) | w(0) | x(1) | w(1)
#pragma unroll
      for (int i = 0; i < MT; ++i) gload16(xr[i], xsrc(i, 0));
#pragma unroll
      for (int sl = 0; sl < SL; ++sl)
#pragma unroll
        for (int j = 0; j < NTW; ++j) gload16_nt(w[sl][j], wbase[j] + (long)sl * 64);
      // x(0) -> LDS needs only x(0): SL*NTW younger loads
      wait_vm<SL * NTW>();
#pragma unroll
      for (int i = 0; i < MT; ++i) st16(smem + xdst0 + i * 4096, xr[i]);
      if (UNIFORM || n > 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i) gload16(xr[i], xsrc(i, UNIFORM ? min(1, n - 1) : 1));
#pragma unroll
        for (int sl = 0; sl < SL; ++sl)
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            gload16_nt(w[SL + sl][j], sel_ptr(!UNIFORM || n > 1, wbase[j] + (long)(SL + sl) * 64, dummy));
      }
    } else {
      // DMA variant: chunk c issues x(c+1) (into the LDS stage chunk c-1 just released) and then
      // the refills w(c+2); the global issue order is w(0) x(0) w(1) | x(1) w(2) | x(2) w(3) ...,
      // which gives the same two wait counts as above with "x(c+1) landed" checked at the END of
      // chunk c, before the barrier that lets chunk c+1 read it.
#pragma unroll
      for (int sl = 0; sl < SL; ++sl)
#pragma unroll
        for (int j = 0; j < NTW; ++j) gload16_nt(w[sl][j], wbase[j] + (long)sl * 64);
      xdma(0, 0);
      if (UNIFORM || n > 1) {
#pragma unroll
        for (int sl = 0; sl < SL; ++sl)
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            gload16_nt(w[SL + sl][j], sel_ptr(n > 1, wbase[j] + (long)(SL + sl) * 64, dummy));
        wait_vm<SL * NTW>();                       // x(0) in LDS; w(1) may still fly
      } else {
        wait_vm<0>();
      }
    }
    __syncthreads();

    // One chunk. REFILL: chunk c+2 exists (steady state, counted waits); otherwise this is one of
    // the last two chunks: nothing new is issued and everything outstanding is drained first.
    auto body = [&](auto PAR, auto REFILL, int c) {
      constexpr int par = decltype(PAR)::value;
      constexpr bool refill = decltype(REFILL)::value != 0;
      if constexpr (XDMA) {
        if constexpr (refill) {
          xdma(par ^ 1, UNIFORM ? min(c + 1, n - 1) : c + 1);
        } else if (c + 1 < n) {
          xdma(par ^ 1, c + 1);
          wait_vm<XL>();                           // everything but the DMA just issued
        } else {
          wait_vm<0>();
        }
      } else if constexpr (refill) {
        char* xn = smem + (par ^ 1) * kStageBytes;
        wait_vm<kWaitX>();
#pragma unroll
        for (int i = 0; i < MT; ++i) st16(xn + xdst0 + i * 4096, xr[i]);
#pragma unroll
        for (int i = 0; i < MT; ++i) gload16(xr[i], xsrc(i, UNIFORM ? min(c + 2, n - 1) : c + 2));
      } else {
        wait_vm<0>();
        if (c + 1 < n) {
          char* xn = smem + (par ^ 1) * kStageBytes;
#pragma unroll
          for (int i = 0; i < MT; ++i) st16(xn + xdst0 + i * 4096, xr[i]);
        }
      }
      const char* xb = smem + par * kStageBytes;
      auto xfrag = [&](int i, int kk) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(xb + (i * 16 + l15) * ROWB + (((kk * 4 + quad) ^ swz(l15)) << 4));
      };
      if constexpr (XDMA && OCC >= 2 && !W8) {
        // Two waves per SIMD, 128 accumulator registers: the x fragments are walked in groups of
        // four m-tiles, double-buffered — the next group's LDS reads are issued before the current
        // group's MFMAs, across k-step and ring-slot boundaries — and a scheduling barrier per group
        // keeps hipcc from hoisting more fragments than the register budget holds.
        constexpr int GPK = MT / 4;                         // groups per k-step
        constexpr int G = KSTEPS * GPK;
        u32x4 xg[2][4];
        u32x4 wf[KPS][NTW];
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) xg[0][i4] = xfrag(i4, 0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int kk = g / GPK, ig = g % GPK;
          if (g + 1 < G) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) xg[(g + 1) & 1][i4] = xfrag(((g + 1) % GPK) * 4 + i4, (g + 1) / GPK);
          }
          if (ig == 0 && kk % KPS == 0) {                   // first use of ring slot kk / KPS
            if constexpr (refill) wait_vm<kWaitW>();
#pragma unroll
            for (int j = 0; j < NTW; ++j) wf[0][j] = w[par * SL + kk / KPS][j];
          }
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
              acc[ig * 4 + i4][j] = mfma16(wf[kk % KPS][j], xg[g & 1][i4], acc[ig * 4 + i4][j]);
          if (ig == GPK - 1 && kk % KPS == KPS - 1) {       // last use of the slot: refill it
            if constexpr (refill) {
#pragma unroll
              for (int j = 0; j < NTW; ++j)
                gload16_nt(w[par * SL + kk / KPS][j],
                           sel_ptr(!UNIFORM || c + 2 < n, wbase[j] + (long)((c + 2) * SL + kk / KPS) * 64, dummy));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (XDMA && !W8 && OCC == 1) {
        // One wave per SIMD here (128 KiB of LDS per workgroup), so nothing else hides the LDS
        // latency: the x fragments of k-step kk+1 are fetched while the MFMAs of kk run. (Not for
        // fp8 weights: with the widened fragments on top, the register file no longer fits both
        // fragment sets and hipcc starts copying ring registers that are still in flight.)
        u32x4 xa[MT], xb2[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xa[i] = xfrag(i, 0);
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) {
          if constexpr (refill) wait_vm<kWaitW>();
          u32x4 wf[KPS][NTW];
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            const u32x4 raw = w[par * SL + sl][j];
            if constexpr (W8) {
              wf[0][j] = fp8x8_to_bf16x8(raw[0], raw[1]);
              wf[KPS - 1][j] = fp8x8_to_bf16x8(raw[2], raw[3]);
            } else {
              wf[0][j] = raw;
            }
          }
          if constexpr (refill && W8) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) gload16_nt(w[par * SL + sl][j], wbase[j] + (long)((c + 2) * SL + sl) * 64);
          }
#pragma unroll
          for (int ks2 = 0; ks2 < KPS; ++ks2) {
            const int kk = sl * KPS + ks2;
            u32x4(&cur)[MT] = (kk & 1) ? xb2 : xa;
            u32x4(&nxt)[MT] = (kk & 1) ? xa : xb2;
            if (kk + 1 < KSTEPS) {
#pragma unroll
              for (int i = 0; i < MT; ++i) nxt[i] = xfrag(i, kk + 1);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NTW; ++j) acc[i][j] = mfma16(wf[ks2][j], cur[i], acc[i][j]);
          }
          if constexpr (refill && !W8) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) gload16_nt(w[par * SL + sl][j], wbase[j] + (long)((c + 2) * SL + sl) * 64);
          }
        }
      } else {
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        if constexpr (refill) wait_vm<kWaitW>();
        u32x4 wf[KPS][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const u32x4 raw = w[par * SL + sl][j];
          if constexpr (W8) {
            wf[0][j] = fp8x8_to_bf16x8(raw[0], raw[1]);
            wf[KPS - 1][j] = fp8x8_to_bf16x8(raw[2], raw[3]);
          } else {
            wf[0][j] = raw;
          }
        }
        if constexpr (refill && W8) {             // the raw slot is free once widened
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            gload16_nt(w[par * SL + sl][j], sel_ptr(!UNIFORM || c + 2 < n, wbase[j] + (long)((c + 2) * SL + sl) * 64, dummy));
        }
#pragma unroll
        for (int ks2 = 0; ks2 < KPS; ++ks2) {
          const int kk = sl * KPS + ks2;
          if constexpr (XDMA || (OCC >= 2 && MT >= 4)) {
            // 256 rows at two waves per SIMD: 128 accumulator registers leave ~100 for everything
            // else, so fragments are fetched four m-tiles at a time and a scheduling barrier per
            // group stops hipcc from hoisting a whole k-step of them (which spills); the LDS
            // latency of a group is covered by the other workgroup on the CU.
#pragma unroll
            for (int ig = 0; ig < MT / 4; ++ig) {
              u32x4 xf[4];
#pragma unroll
              for (int i4 = 0; i4 < 4; ++i4) xf[i4] = xfrag(ig * 4 + i4, kk);
#pragma unroll
              for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[ig * 4 + i4][j] = mfma16(wf[ks2][j], xf[i4], acc[ig * 4 + i4][j]);
              __builtin_amdgcn_sched_barrier(0);
            }
          } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const u32x4 xf = xfrag(i, kk);
#pragma unroll
              for (int j = 0; j < NTW; ++j) acc[i][j] = mfma16(wf[ks2][j], xf, acc[i][j]);
            }
          }
        }
        if constexpr (refill && !W8) {
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            gload16_nt(w[par * SL + sl][j], sel_ptr(!UNIFORM || c + 2 < n, wbase[j] + (long)((c + 2) * SL + sl) * 64, dummy));
        }
      }
      }
      if constexpr (XDMA) {                        // x(c+1) must be in LDS before anyone passes the barrier
        if constexpr (refill) wait_vm<kWaitX>();
        else wait_vm<0>();
      }
      __syncthreads();
    };
    int c = 0;
    if constexpr (UNIFORM) {
      for (; c < n; c += 2) {
        body(IntC<0>{}, IntC<1>{}, c);
        if (c + 1 < n) body(IntC<1>{}, IntC<1>{}, c + 1);
      }
      wait_vm<0>();                               // the dummy loads must land before their registers are reused
      c = n;
    }
    for (; c + 3 < n; c += 2) {                  // both chunks of the pair have a chunk two ahead
      body(IntC<0>{}, IntC<1>{}, c);
      body(IntC<1>{}, IntC<1>{}, c + 1);
    }
    if (UNIFORM) {
    } else if (n - c == 3) {
      body(IntC<0>{}, IntC<1>{}, c);
      body(IntC<1>{}, IntC<0>{}, c + 1);
      body(IntC<0>{}, IntC<0>{}, c + 2);
    } else if (n - c == 2) {
      body(IntC<0>{}, IntC<0>{}, c);
      body(IntC<1>{}, IntC<0>{}, c + 1);
    } else {
      body(IntC<0>{}, IntC<0>{}, c);
    }
  }
  if (!wave_active) return;

  // acc[i][j][r] = C[m = i*16 + l15][n = (tile0+j)*16 + quad*4 + r]
  if constexpr (W8) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(p.wscale + (tile0 + j) * 16 + quad * 4);
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] *= sc;
    }
  }
  if (p.partial_only || p.splitk > 1) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = i * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int n = (tile0 + j) * 16 + quad * 4;
        if (n < p.N)
          *reinterpret_cast<f32x4*>(p.partial + ((long)ks * p.M + m) * p.N + n) = acc[i][j];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = i * 16 + l15;
    if (m >= p.M) continue;
    if (p.act == 1) {
      // tiles (2t, 2t+1) = (gate, up) rows of output columns [16t, 16t+16)
      if constexpr (NTW >= 2) {
#pragma unroll
        for (int j = 0; j < NTW; j += 2) {
          const int n = ((tile0 + j) >> 1) * 16 + quad * 4;
          if (n < (p.N >> 1)) {
            float g[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const float u[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
            epilogue4(p, m, n, g, u);
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int n = (tile0 + j) * 16 + quad * 4;
        if (n < p.N) {
          float g[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          epilogue4(p, m, n, g, g);
        }
      }
    }
  }
  // As the lm_head (plain fp32 output, no split): the wave's two n-tiles are one 32-column chunk of the logits
  // row; its maximum over the columns below n_valid goes to chunk_max[m][chunk] for the guided sampler.
  if constexpr (NTW == 2) {
    if (p.chunk_max != nullptr) {
      const int chunk = tile0 >> 1;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float mx = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((tile0 + j) * 16 + quad * 4 + r < p.n_valid) mx = fmaxf(mx, acc[i][j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const int m = i * 16 + l15;
        if (quad == 0 && m < p.M) p.chunk_max[(long)m * p.ldc + chunk] = mx;
      }
    }
  }
}

// Sum S fp32 partials [S][M][N] and apply the epilogue. One thread per 4 output columns.
// For swiglu the partial columns are tile-interleaved: gate of output column c lives at GEMM
// column (c/16)*32 + c%16, up at +16.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
  const int nout = p.act == 1 ? p.N >> 1 : p.N;
  const int per_row = nout >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.M * per_row) return;
  const int m = (int)(idx / per_row);
  const int n = (int)(idx - (long)m * per_row) * 4;
  float g[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0};
  const int gn = p.act == 1 ? (n >> 4) * 32 + (n & 15) : n;
  for (int s = 0; s < p.splitk; ++s) {
    const float* row = p.partial + ((long)s * p.M + m) * p.N;
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + gn);
    g[0] += a[0]; g[1] += a[1]; g[2] += a[2]; g[3] += a[3];
    if (p.act == 1) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(row + gn + 16);
      u[0] += b[0]; u[1] += b[1]; u[2] += b[2]; u[3] += b[3];
    }
  }
  epilogue4(p, m, n, g, u);
}

template <int MT, int NTW, int BK, int OCC>
static void launch_skinny(const GemmParams& p, hipStream_t stream) {
  const int nblocks = (p.ntiles + 4 * NTW - 1) / (4 * NTW);
  dim3 grid(nblocks, p.splitk), block(kSkinnyThreads);
  constexpr int smem = 2 * MT * 16 * BK * 2;
  if constexpr (smem > 64 * 1024) {               // more than 64 KiB of dynamic LDS is opt-in, once per kernel
    static const bool once = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<MT, NTW, true, BK, OCC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<MT, NTW, false, BK, OCC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      return true;
    }();
    (void)once;
  }
  if (p.wscale) hipLaunchKernelGGL((gemm_skinny_kernel<MT, NTW, true, BK, OCC>), grid, block, smem, stream, p);
  else          hipLaunchKernelGGL((gemm_skinny_kernel<MT, NTW, false, BK, OCC>), grid, block, smem, stream, p);
}

template <int MT>
static void launch_skinny_ntw(const GemmParams& p, int ntw, hipStream_t stream) {
  if constexpr (MT == 16) {
    static const bool one_per_cu = [] { const char* e = getenv("DLIA_GEMM_256_ONE_PER_CU"); return e && atoi(e) != 0; }();
    if (one_per_cu) {                             // 128-wide chunks, one workgroup per CU (kept for A/B runs)
      if (ntw == 1) launch_skinny<MT, 1, 128, 1>(p, stream);
      else          launch_skinny<MT, 2, 128, 1>(p, stream);
    } else {
      if (ntw == 1) launch_skinny<MT, 1, 64, 2>(p, stream);
      else          launch_skinny<MT, 2, 64, 2>(p, stream);
    }
  } else {
    if (ntw == 1) launch_skinny<MT, 1, 128, 1>(p, stream);
    else          launch_skinny<MT, 2, 128, 1>(p, stream);
  }
}

}  // namespace dlia

using namespace dlia;

// Decomposition choice (calibrated with bench/gemm_sweep.py on MI355X). A workgroup streams 128
// weight columns (2 n-tiles per wave): that re-reads x least. Shapes with fewer than 256 column
// blocks are split along K until every CU has a workgroup — measured optimum for the Llama-3-8B
// decode shapes (QKV 48 blocks -> 8 splits, O 32 -> 8, down 32 -> 8, gate/up 224 -> 2, lm_head
// 1002 -> 1) — as long as a split keeps >= 2 LDS stages and the fp32 partials stay under ~20 MB
// (at 128 rows QKV drops to 4 splits and gate/up to 1, also the measured optimum).
extern "C" void dlia_gemm_skinny_config(int M, int N, int K, int epi, int allow_split, int* ntw_out, int* splitk_out) {
  const int tiles = packed_ntiles(N);
  const int chunks = K / kSkinnyBK;
  const int ntw = 2;
  const int nblocks = (tiles + 4 * ntw - 1) / (4 * ntw);
  int s = 1;
  if (allow_split) {
    // split until ~3/4 of the 256 CUs have a workgroup: a launch of 224 workgroups streams faster
    // unsplit than as 448 half-length ones that also write twice the partials
    static const int forced = [] {                // tuning override: workgroups a launch should reach
      const char* e = getenv("DLIA_GEMM_TARGET_WGS");
      return e ? atoi(e) : 0;
    }();
    // Above 128 rows a launch is no longer a pure weight stream and the serving path overlaps two or
    // three micro-batches on separate streams: a narrower split (fewer, longer workgroups, fewer
    // fp32 partials) leaves CUs to the other streams. Measured at 256 rows, target 128 vs 192:
    // 3 x 256 sequences on 3 streams 45.2k vs 42.5k tok/s, a single-stream step 7.43 vs 7.32 ms.
    const int target = forced > 0 ? forced : (M > 128 ? 128 : 192);
    while (nblocks * s < target && s < 16 && chunks / (s * 2) >= 2) s *= 2;
    const double cap = M > 128 ? 36.0e6 : 20.0e6;                  // cap the fp32 partial traffic
    while (s > 1 && (double)s * M * N * 4.0 > cap) s /= 2;
    // A fused epilogue (bias / residual / activation: anything but "partials only") needs a reduce
    // launch after a split, ~5 us of a decode step that is a chain of ~5 us kernels. From 64
    // workgroups on, the unsplit launch is faster than split + reduce: TinyLlama gate/up (88
    // workgroups unsplit, 4 splits before) 1.155 -> 1.113 ms per token at batch 1, 1.66 -> 1.54 at 64;
    // GPT-2's fc (24 workgroups) keeps its split (0.507 against 0.518 ms unsplit).
    if (epi != 6 && nblocks >= 64) s = 1;
  }
  *ntw_out = ntw; *splitk_out = s;
}

// Sum the split-K partials of `p` and apply its epilogue (used by every streaming GEMM variant).
extern "C" void dlia_splitk_reduce(const dlia::GemmParams* p, hipStream_t stream) {
  const int nout = p->act == 1 ? p->N / 2 : p->N;
  const long work = (long)p->M * (nout / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, *p);
}

extern "C" int dlia_gemm_skinny_splitk(int M, int N, int K) {
  int ntw, s;
  dlia_gemm_skinny_config(M, N, K, 0, 1, &ntw, &s);
  return s;
}

// epi: 0..5 as documented in launch.h; 6 = write fp32 partials [splitk][M][N] only (the consumer
// kernel sums them). splitk / ntw <= 0 pick the heuristic.
extern "C" void dlia_gemm_skinny(void* out, const void* x, const void* wp, const void* bias,
                                 const void* residual, float* workspace, const float* wscale,
                                 int M, int N, int K, long ldx, long ldo, long ldr, int epi,
                                 int out_f32, int splitk, int ntw, hipStream_t stream) {
  if (M <= 0 || N <= 0) return;
  GemmParams p;
  p.counters = nullptr; p.group_n = 1;
  p.x = (const bf16*)x; p.ldx = ldx; p.wp = (const u32x4*)wp; p.wscale = wscale; p.xscale = nullptr; p.out = out; p.ldo = ldo;
  p.bias = (epi == 1 || epi == 3 || epi == 5) ? (const bf16*)bias : nullptr;
  p.residual = (epi == 2 || epi == 3) ? (const bf16*)residual : nullptr;
  p.ldr = ldr; p.partial = workspace; p.M = M; p.N = N; p.K = K;
  p.ntiles = packed_ntiles(N);
  p.act = epi == 4 ? 1 : (epi == 5 ? 2 : 0);
  p.out_f32 = out_f32;
  p.partial_only = epi == 6 ? 1 : 0;
  int antw, as;
  dlia_gemm_skinny_config(M, N, K, epi, workspace != nullptr, &antw, &as);
  if (splitk <= 0) splitk = as;
  if (ntw <= 0) ntw = antw;
  if (epi == 4) ntw = 2;
  if (workspace == nullptr) splitk = 1;
  const int chunks = K / kSkinnyBK;
  if (splitk > chunks) splitk = chunks;
  p.splitk = splitk;
  // bf16 above 128 rows: the all-DMA kernel with loader waves (gemm_rows.hip). Up to 128 rows the
  // register-ring variants here already stream at the HBM ceiling; fp8 weights use them at every
  // row count (they widen the weights in registers). DLIA_GEMM_RING=1 selects them for bf16 at
  // 129..256 rows as well (A/B runs).
  static const bool ring = [] { const char* e = getenv("DLIA_GEMM_RING"); return e && atoi(e) != 0; }();
  if (!wscale && M > 128 && !(ring && M <= 256)) {
    dlia_gemm_rows(&p, stream);
  } else if (M <= 16)  launch_skinny_ntw<1>(p, ntw, stream);
  else if (M <= 32)  launch_skinny_ntw<2>(p, ntw, stream);
  else if (M <= 64)  launch_skinny_ntw<4>(p, ntw, stream);
  else if (M <= 128) launch_skinny_ntw<8>(p, ntw, stream);
  else               launch_skinny_ntw<16>(p, ntw, stream);
  if (splitk > 1 && !p.partial_only) dlia_splitk_reduce(&p, stream);
}

// lm_head for at most 256 rows (up to 128: this file's kernel; 129..256: gemm_rows.hip): logits[M, N] (fp32) = x @ Wp^T
// with no split, and chunk_max[m * ldc + n / 32] = the
// maximum of every 32-column chunk of every row over the columns below n_valid (the padding columns of the packed
// weight would otherwise compete with real logits), for every chunk of the packed width (N rounded up to 128).
extern "C" void dlia_gemm_skinny_head(void* logits, const void* x, const void* wp, int M, int N, int K, long ldx, long ldo,
                                      float* chunk_max, long ldc, int n_valid, hipStream_t stream) {
  if (M <= 0 || N <= 0 || M > 256) return;
  GemmParams p;
  p.counters = nullptr; p.group_n = 1;
  p.x = (const bf16*)x; p.ldx = ldx; p.wp = (const u32x4*)wp; p.wscale = nullptr; p.xscale = nullptr; p.out = logits; p.ldo = ldo;
  p.bias = nullptr; p.residual = nullptr; p.ldr = 0; p.partial = nullptr; p.M = M; p.N = N; p.K = K;
  p.ntiles = packed_ntiles(N);
  p.act = 0; p.out_f32 = 1; p.partial_only = 0; p.splitk = 1;
  p.chunk_max = chunk_max; p.ldc = ldc; p.n_valid = n_valid;
  if (M > 128)      dlia_gemm_rows(&p, stream);
  else if (M <= 16) launch_skinny_ntw<1>(p, 2, stream);
  else if (M <= 32) launch_skinny_ntw<2>(p, 2, stream);
  else if (M <= 64) launch_skinny_ntw<4>(p, 2, stream);
  else              launch_skinny_ntw<8>(p, 2, stream);
}
