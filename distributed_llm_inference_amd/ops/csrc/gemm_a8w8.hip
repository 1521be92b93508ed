// fp8 x fp8 weight-streaming GEMM on the CDNA4 block-scaled matrix instruction:
// out[M,N] = epi( (xq[M,K] * xscale[M]) @ (Wq[N,K] * wscale[N])^T ), M <= 256.
//
// Both operands are OCP e4m3. v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales consumes
// 128 K-elements per instruction at twice the per-clock rate of the bf16 form, and an fp8
// activation row is half the LDS bytes of a bf16 one — the two resources that bound the bf16
// kernel at 128-256 rows. Dequantisation is exact and free: one per-row and one per-column fp32
// scale on the accumulators before the epilogue.
//
// Structure = the 64-wide-chunk bf16 streaming kernel (gemm_skinny.hip): 4 waves x NTW n-tiles,
// a chunk is 128 K-elements = 128-byte x rows in LDS (two stages, staged by LDS-DMA with the XOR
// swizzle on the source column), weights straight to VGPRs through a two-chunk register ring with
// hand-counted vmcnt, ONE loop body per chunk parity (loads past the end of a split go to a dummy
// address so the counts hold to the last chunk), two workgroups per CU.
//
// Operand layout (verified with exact integer data): lane l (r = l & 15, q = l >> 4) byte j of
// its 32-byte fragment is element [row r][k = 32 q + j] of the 16 x 128 operand; C/D as for bf16.
// Packed weights (ops/packing.py::pack_weight_a8w8): [n-tile][K/128][half 0|1][64 lanes][16 B],
// so each of the two ring loads of a fragment is one contiguous 1 KiB wave access.
//
// Replaces, for fp8 deployments, the nn.Linear calls the reference runs in fp32 on CPU
// (Worker1.py:128-137; orchestration.py:141).
#include <cstdlib>
#include "common.h"
#include "gemm_common.h"
#include "launch.h"

namespace dlia {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int kA8Threads = 256;

template <int V> struct IntA { static constexpr int value = V; };

__device__ __forceinline__ void a8_gload16_nt(u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p));
}
template <int N>
__device__ __forceinline__ void a8_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ const void* a8_sel_ptr(bool c, const void* a, const void* b) {
  const long d = reinterpret_cast<const char*>(a) - reinterpret_cast<const char*>(b);
  return reinterpret_cast<const char*>(b) + (d & -static_cast<long>(c));
}

// MT in {4, 8, 16} (64 / 128 / 256 rows), NTW in {1, 2}, KSUB = MFMA k-steps (128 elements each)
// per chunk: 2 up to 128 rows — twice the weight bytes in flight per wave, which is what a
// memory-bound launch needs — and 1 at 256 rows, where two 64 KiB x stages would leave room for
// only one workgroup per CU.
template <int MT, int NTW, int KSUB>
__global__ __launch_bounds__(kA8Threads, 2) void gemm_a8w8_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages of MT*16 rows x ROWB bytes
  constexpr int ROWB = 128 * KSUB;
  constexpr int BKE = 128 * KSUB;                // K elements per chunk
  constexpr int kStageBytes = MT * 16 * ROWB;
  constexpr int RPI = 1024 / ROWB;               // rows per DMA instruction
  constexpr int LPR = ROWB / 16;                 // 16-B slots per row
  constexpr int XL = MT * 16 / RPI / 4;          // DMA instructions per wave and chunk
  constexpr int HL = 2 * KSUB;                   // 16-B ring loads per n-tile and chunk
  static_assert(XL >= 1, "at least 64 rows");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int nblk = blockIdx.x, ks = blockIdx.y;
  const int KC = p.K / BKE;                      // chunks in K
  const int cper = (KC + p.splitk - 1) / p.splitk;
  const int c0 = ks * cper;
  const int c1 = min(KC, c0 + cper);
  const int n = c1 - c0;
  const int tile0 = (nblk * 4 + wave) * NTW;
  const bool wave_active = tile0 < p.ntiles;
  auto swz = [](int row) { return KSUB == 1 ? ((row >> 1) & 7) : (row & 15); };

  const unsigned char* xq = reinterpret_cast<const unsigned char*>(p.x);
  const void* const dummy = p.x;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  auto xdma = [&](int stage, int c) {
    const void* src[XL];
#pragma unroll
    for (int q = 0; q < XL; ++q) {
      const int row = RPI * (wv * XL + q) + lane / LPR;
      src[q] = xq + (long)(row < p.M ? row : p.M - 1) * p.ldx + (((lane % LPR) ^ swz(row)) << 4) + (long)(c0 + c) * BKE;
    }
    lds_dma_pieces<XL>(src, lds0 + stage * kStageBytes + wv * XL * 1024);
  };
  const u32x4* wbase[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
    wbase[j] = p.wp + ((long)(wave_active ? tile0 + j : 0) * KC + c0) * HL * 64 + lane;

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (n > 0) {
    // Issue order: w(0) x(0) w(1) | x(1) w(2) | x(2) w(3) | ...  (chunk c issues x(c+1), then,
    // after its MFMAs, the refill w(c+2)). Both halves of a chunk's weight fragment are consumed
    // together, so the wait is on the younger half:
    //   w(c) has x(c), w(c+1), x(c+1) younger at its use      -> vmcnt(2 XL + 2 NTW)
    //   x(c+1) has w(c+2) younger at the end of chunk c        -> vmcnt(2 NTW)
    constexpr int kWaitW = 2 * XL + HL * NTW, kWaitX = HL * NTW;     // (HL = 2 halves x KSUB k-steps)
    static_assert(kWaitW <= 63, "vmcnt is a 6-bit counter");
    u32x4 w[2][HL][NTW];                           // [chunk parity][k-step * 2 + half][n-tile]
#pragma unroll
    for (int h = 0; h < HL; ++h)
#pragma unroll
      for (int j = 0; j < NTW; ++j) a8_gload16_nt(w[0][h][j], wbase[j] + (long)h * 64);
    xdma(0, 0);
#pragma unroll
    for (int h = 0; h < HL; ++h)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        a8_gload16_nt(w[1][h][j], a8_sel_ptr(n > 1, wbase[j] + (long)(HL + h) * 64, dummy));
    a8_wait_vm<kWaitX>();                          // x(0) in LDS; w(1) may still fly
    __syncthreads();

    auto body = [&](auto PAR, int c) {
      constexpr int par = decltype(PAR)::value;
      xdma(par ^ 1, min(c + 1, n - 1));
      const char* xb = smem + par * kStageBytes;
      a8_wait_vm<kWaitW>();
#pragma unroll
      for (int kq = 0; kq < KSUB; ++kq) {
        i32x8 wf[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const u32x4 lo = w[par][2 * kq][j], hi = w[par][2 * kq + 1][j];
          wf[j] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int ig = 0; ig < MT / 4; ++ig) {
          i32x8 xf[4];
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) {
            const char* rowp = xb + ((ig * 4 + i4) * 16 + l15) * ROWB;
            const u32x4 lo = *reinterpret_cast<const u32x4*>(rowp + (((kq * 8 + 2 * quad) ^ swz(l15)) << 4));
            const u32x4 hi = *reinterpret_cast<const u32x4*>(rowp + (((kq * 8 + 2 * quad + 1) ^ swz(l15)) << 4));
            xf[i4] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
          }
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
              acc[ig * 4 + i4][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                  wf[j], xf[i4], acc[ig * 4 + i4][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int h = 0; h < HL; ++h)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          a8_gload16_nt(w[par][h][j], a8_sel_ptr(c + 2 < n, wbase[j] + (long)((c + 2) * HL + h) * 64, dummy));
      a8_wait_vm<kWaitX>();                        // x(c+1) must be in LDS before anyone passes the barrier
      __syncthreads();
    };
    for (int c = 0; c < n; c += 2) {
      body(IntA<0>{}, c);
      if (c + 1 < n) body(IntA<1>{}, c + 1);
    }
    a8_wait_vm<0>();                               // the dummy loads must land before their registers are reused
  }
  if (!wave_active) return;

  // dequantise: per output column (weight row) and per row (token)
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.wscale + (tile0 + j) * 16 + quad * 4);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i][j] *= sc;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = i * 16 + l15;
    const float xs = p.xscale[m < p.M ? m : p.M - 1];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] *= xs;
  }
  write_acc_tiles<MT, NTW>(p, acc, tile0, l15, quad, ks);
}

// Per-row dynamic quantisation of activations: q = x / s, s = max|x| / 448. One workgroup per row.
__global__ __launch_bounds__(256) void quant_rows_kernel(unsigned char* __restrict__ q, float* __restrict__ scale,
                                                         const bf16* __restrict__ x, int K, long ldx, long ldq) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16* xr = x + (long)row * ldx;
  const int nvec = K >> 3;
  float amax = 0.f;
  for (int i = tid; i < nvec; i += 256) {
    float f[8];
    unpack8(ld16(xr + i * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float s = fmaxf(amax, 1e-12f) * (1.f / 448.f);
  const float inv = 1.f / s;
  if (tid == 0) scale[row] = s;
  for (int i = tid; i < nvec; i += 256) {
    float f[8];
    unpack8(ld16(xr + i * 8), f);
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
    *reinterpret_cast<u32x2*>(q + (long)row * ldq + i * 8) = u32x2{lo, hi};
  }
}

// 256-element chunks up to 128 rows when K allows it, 128-element chunks otherwise
static inline int a8w8_ksub(int M, int K) { return (M <= 128 && K % 256 == 0) ? 2 : 1; }

template <int MT, int NTW, int KSUB>
static void launch_a8w8_k(const GemmParams& p, hipStream_t stream) {
  const int nblocks = (p.ntiles + 4 * NTW - 1) / (4 * NTW);
  dim3 grid(nblocks, p.splitk), block(kA8Threads);
  constexpr int smem = 2 * MT * 16 * 128 * KSUB;
  static_assert(smem <= 64 * 1024, "stays within the default dynamic LDS limit");
  hipLaunchKernelGGL((gemm_a8w8_kernel<MT, NTW, KSUB>), grid, block, smem, stream, p);
}

template <int MT, int NTW>
static void launch_a8w8(const GemmParams& p, hipStream_t stream) {
  if constexpr (MT < 16) {
    if (a8w8_ksub(p.M, p.K) == 2) { launch_a8w8_k<MT, NTW, 2>(p, stream); return; }
  }
  launch_a8w8_k<MT, NTW, 1>(p, stream);
}

template <int MT>
static void launch_a8w8_ntw(const GemmParams& p, int ntw, hipStream_t stream) {
  if (ntw == 1) launch_a8w8<MT, 1>(p, stream);
  else          launch_a8w8<MT, 2>(p, stream);
}

}  // namespace dlia

using namespace dlia;

extern "C" void dlia_quant_rows(void* q, float* scale, const void* x, int M, int K, long ldx, long ldq,
                                hipStream_t stream) {
  if (M <= 0) return;
  hipLaunchKernelGGL(quant_rows_kernel, dim3(M), dim3(256), 0, stream, (unsigned char*)q, scale, (const bf16*)x, K,
                     ldx, ldq);
}

// Same contract as dlia_gemm_skinny, with xq = e4m3 bytes [M, K] (row stride ldx bytes), xscale[M],
// wp from pack_weight_a8w8 and wscale per packed row. Split-K (workspace) only up to 256 rows.
extern "C" void dlia_gemm_a8w8(void* out, const void* xq, const float* xscale, const void* wp, const float* wscale,
                               const void* bias, const void* residual, float* workspace, int M, int N, int K,
                               long ldx, long ldo, long ldr, int epi, int out_f32, int splitk, int ntw,
                               hipStream_t stream) {
  if (M <= 0 || N <= 0) return;
  GemmParams p;
  p.counters = nullptr; p.group_n = 1;
  p.x = (const bf16*)xq; p.ldx = ldx; p.wp = (const u32x4*)wp; p.wscale = wscale; p.xscale = xscale;
  p.out = out; p.ldo = ldo;
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
  if (ntw <= 0 || ntw > 2) ntw = 2;
  if (epi == 4) ntw = 2;
  if (workspace == nullptr) splitk = 1;
  const int chunks = K / (128 * a8w8_ksub(M, K));
  if (splitk > chunks) splitk = chunks;
  p.splitk = splitk;
  // above 128 rows: the all-DMA kernel with loader waves (gemm_rows.hip, fp8 variant), which also
  // walks 256-row blocks for M > 256; DLIA_GEMM_RING=1 keeps the register-ring variant at 129..256
  static const bool ring = [] { const char* e = getenv("DLIA_GEMM_RING"); return e && atoi(e) != 0; }();
  if (M > 256 || (M > 128 && !ring)) dlia_gemm_rows_f8(&p, stream);
  else if (M <= 64)  launch_a8w8_ntw<4>(p, ntw, stream);
  else if (M <= 128) launch_a8w8_ntw<8>(p, ntw, stream);
  else               launch_a8w8_ntw<16>(p, ntw, stream);
  if (splitk > 1 && !p.partial_only) dlia_splitk_reduce(&p, stream);
}
