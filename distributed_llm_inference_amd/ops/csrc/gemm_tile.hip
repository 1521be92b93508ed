// Tiled MFMA GEMM for the prefill regime (M in the hundreds..thousands):
// out[M,N] = epi(x[M,K] @ W[N,K]^T) on 128x128x64 tiles, 4 waves (2x2), each wave 64x64.
//
//  * Both operands are staged global -> LDS with `global_load_lds` (16 B per lane, no VGPR
//    round trip), double-buffered: the loads of K-tile t+1 are issued before the MFMAs of
//    K-tile t and retired by one vmcnt(0) + barrier per K-tile.
//  * W comes from the same packed-B image the decode kernel streams: a (n-tile, k-step) block
//    is already 64 lanes x 16 B in fragment order, so its LDS image is lane-linear and its
//    ds_read_b128 fragment reads are bank-conflict-free with no swizzle.
//  * x tiles are 128-B rows; the LDS-DMA destination must stay lane-linear, so the XOR swizzle
//    (16-B slot ^= (row>>1)&7) is applied to the per-lane SOURCE address and again on the read.
//  * MFMA roles swapped (A = W, B = x) as in the decode kernel: 4 consecutive columns per lane.
//  * blockIdx is remapped so that the 8 workgroups sharing an XCD's L2 walk neighbouring tiles.
#include "common.h"
#include "gemm_common.h"
#include "launch.h"

namespace dlia {

constexpr int kTileM = 128, kTileN = 128, kTileK = 64;
constexpr int kTileThreads = 256;
constexpr int kTileStageBytes = kTileM * kTileK * 2 + kTileN * kTileK * 2;   // 32 KiB

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(kTileThreads) void gemm_tile_kernel(GemmParams p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  // consecutive logical ids walk M first: neighbours share the W panel (the big operand)
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * kTileM;
  const int ntile0 = tn * (kTileN / 16);
  const int KT = p.K >> 5;
  const int nk = p.K / kTileK;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging addresses (per lane, constant across K-tiles except for the k offset) ----
  const bf16* xsrc[4];
  const u32x4* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;                 // 16 pieces of 1 KiB per operand tile
    const int row = piece * 8 + (lane >> 3);        // x: 8 rows x 128 B per piece
    const int phys = lane & 7;
    const int logical = phys ^ ((row >> 1) & 7);
    const int gr = min(m0 + row, p.M - 1);
    xsrc[i] = p.x + (long)gr * p.ldx + logical * 8;
    const int ntl = piece >> 1, kk = piece & 1;     // W: block (n-tile ntl, k-step kk)
    wsrc[i] = p.wp + ((long)(ntile0 + ntl) * KT + kk) * 64 + lane;
  }
  auto stage = [&](int buf, int t) {
    char* a = smem + buf * kTileStageBytes;
    char* b = a + kTileM * kTileK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(xsrc[i] + (long)t * kTileK, a + (wave * 4 + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(wsrc[i] + (long)t * 2 * 64, b + (wave * 4 + i) * 1024);
  };

  f32x4 acc[4][4];   // [n-tile j][m-tile i]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* a = smem + buf * kTileStageBytes;
    const char* b = a + kTileM * kTileK * 2;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 wf[4], xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wf[j] = *reinterpret_cast<const u32x4*>(b + (((wn * 4 + j) * 2 + kk) * 64 + lane) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + l15;
        const int slot = (kk * 4 + quad) ^ ((row >> 1) & 7);
        xf[i] = *reinterpret_cast<const u32x4*>(a + row * 128 + slot * 16);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mfma16(wf[j], xf[i], acc[j][i]);
    }
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nk - 1; ++t) {
    stage(cur ^ 1, t + 1);
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  compute(cur);

  // acc[j][i][r] = C[m0 + wm*64 + i*16 + l15][(ntile0 + wn*4 + j)*16 + quad*4 + r]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l15;
    if (m >= p.M) continue;
    if (p.act == 1) {
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int n = ((ntile0 + wn * 4 + j) >> 1) * 16 + quad * 4;
        if (n < (p.N >> 1)) {
          float g[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
          const float u[4] = {acc[j + 1][i][0], acc[j + 1][i][1], acc[j + 1][i][2], acc[j + 1][i][3]};
          epilogue4(p, m, n, g, u);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = (ntile0 + wn * 4 + j) * 16 + quad * 4;
        if (n < p.N) {
          float g[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
          epilogue4(p, m, n, g, g);
        }
      }
    }
  }
}

}  // namespace dlia

using namespace dlia;

extern "C" void dlia_gemm_tile(void* out, const void* x, const void* wp, const void* bias,
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
  const int tiles_m = (M + kTileM - 1) / kTileM;
  const int tiles_n = p.ntiles / (kTileN / 16);
  hipLaunchKernelGGL(gemm_tile_kernel, dim3(tiles_m * tiles_n), dim3(kTileThreads), 2 * kTileStageBytes,
                     stream, p, tiles_m, tiles_n);
}
