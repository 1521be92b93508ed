// Weight-streaming GEMM for the decode regime: out[M,N] = epi(x[M,K] @ W[N,K]^T), M <= 64.
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
// Replaces the nn.Linear calls inside HF LlamaDecoderLayer that the reference runs on CPU in
// fp32 over the whole sequence each step (Worker1.py:128-137) and the lm_head
// (orchestration.py:141).
#include "common.h"
#include "gemm_common.h"
#include "launch.h"

namespace dlia {

constexpr int kSkinnyBK = 128;       // K elements per LDS stage (4 MFMA k-steps)
constexpr int kSkinnyThreads = 256;  // 4 waves

template <int MT, int NTW>
__global__ __launch_bounds__(kSkinnyThreads) void gemm_skinny_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (MT*16 rows x 256 B)
  constexpr int kStageBytes = MT * 16 * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int nblk = blockIdx.x, ks = blockIdx.y;
  const int KT = p.K >> 5;                       // MFMA k-steps in K
  const int chunks_total = p.K / kSkinnyBK;
  const int cper = (chunks_total + p.splitk - 1) / p.splitk;
  const int c0 = ks * cper;
  const int c1 = min(chunks_total, c0 + cper);
  const int tile0 = (nblk * 4 + wave) * NTW;     // first n-tile of this wave
  const bool wave_active = tile0 < p.ntiles;     // (ntiles is a multiple of 4*NTW by packing)

  // x staging map: thread -> (row = tid/16 + 16 i, 16-B slot = tid%16), slot XOR-swizzled by row&15
  const int srow = tid >> 4, sslot = tid & 15;
  const bf16* xsrc[MT];
  int xdst[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int r = srow + 16 * i;
    const int rc = r < p.M ? r : p.M - 1;        // rows past M re-read the last row; never stored
    xsrc[i] = p.x + (long)rc * p.ldx + sslot * 8;
    xdst[i] = r * 256 + ((sslot ^ (r & 15)) << 4);
  }
  const u32x4* wbase[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
    wbase[j] = p.wp + ((long)(wave_active ? tile0 + j : 0) * KT) * 64 + lane;

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (c0 < c1) {
    u32x4 wcur[4][NTW], wnext[4][NTW], xr[MT];
    // prologue: chunk c0
#pragma unroll
    for (int i = 0; i < MT; ++i) xr[i] = ld16(xsrc[i] + (long)c0 * kSkinnyBK);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < NTW; ++j) wcur[kk][j] = ld16_nt(wbase[j] + (long)(c0 * 4 + kk) * 64);
#pragma unroll
    for (int i = 0; i < MT; ++i) st16(smem + xdst[i], xr[i]);
    __syncthreads();

    for (int c = c0; c < c1; ++c) {
      const int buf = (c - c0) & 1;
      const bool more = c + 1 < c1;
      if (more) {
#pragma unroll
        for (int i = 0; i < MT; ++i) xr[i] = ld16(xsrc[i] + (long)(c + 1) * kSkinnyBK);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int j = 0; j < NTW; ++j) wnext[kk][j] = ld16_nt(wbase[j] + (long)((c + 1) * 4 + kk) * 64);
      }
      const char* xb = smem + buf * kStageBytes;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int r = i * 16 + l15;
          const u32x4 xf = *reinterpret_cast<const u32x4*>(xb + r * 256 + (((kk * 4 + quad) ^ l15) << 4));
#pragma unroll
          for (int j = 0; j < NTW; ++j) acc[i][j] = mfma16(wcur[kk][j], xf, acc[i][j]);
        }
      }
      if (more) {
        char* xn = smem + (buf ^ 1) * kStageBytes;
#pragma unroll
        for (int i = 0; i < MT; ++i) st16(xn + xdst[i], xr[i]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int j = 0; j < NTW; ++j) wcur[kk][j] = wnext[kk][j];
      }
      __syncthreads();
    }
  }
  if (!wave_active) return;

  // acc[i][j][r] = C[m = i*16 + l15][n = (tile0+j)*16 + quad*4 + r]
  if (p.splitk > 1) {
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
      static_assert(NTW % 2 == 0 || NTW == 1, "swiglu needs gate/up tile pairs in one wave");
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

template <int MT, int NTW>
static void launch_skinny(const GemmParams& p, hipStream_t stream) {
  const int nblocks = (p.ntiles + 4 * NTW - 1) / (4 * NTW);
  dim3 grid(nblocks, p.splitk), block(kSkinnyThreads);
  const size_t smem = 2 * MT * 16 * 256;
  hipLaunchKernelGGL((gemm_skinny_kernel<MT, NTW>), grid, block, smem, stream, p);
}

}  // namespace dlia

using namespace dlia;

extern "C" int dlia_gemm_skinny_splitk(int M, int N, int K) {
  // Aim for >= ~512 workgroups (2 per CU) while keeping >= 4 LDS stages (512 K-elements) per split.
  const int ntw = 2;
  const int nblocks = (packed_ntiles(N) + 4 * ntw - 1) / (4 * ntw);
  const int chunks = K / kSkinnyBK;
  int s = 1;
  while (nblocks * s < 384 && chunks / (s * 2) >= 4 && s < 16) s *= 2;
  (void)M;
  return s;
}

extern "C" void dlia_gemm_skinny(void* out, const void* x, const void* wp, const void* bias,
                                 const void* residual, float* workspace, unsigned* counters,
                                 int M, int N, int K, long ldx, long ldo, long ldr, int epi,
                                 int out_f32, int splitk, hipStream_t stream) {
  (void)counters;
  if (M <= 0 || N <= 0) return;
  GemmParams p;
  p.x = (const bf16*)x; p.ldx = ldx; p.wp = (const u32x4*)wp; p.out = out; p.ldo = ldo;
  p.bias = (epi == 1 || epi == 3 || epi == 5) ? (const bf16*)bias : nullptr;
  p.residual = (epi == 2 || epi == 3) ? (const bf16*)residual : nullptr;
  p.ldr = ldr; p.partial = workspace; p.M = M; p.N = N; p.K = K;
  p.ntiles = packed_ntiles(N);
  p.act = epi == 4 ? 1 : (epi == 5 ? 2 : 0);
  p.out_f32 = out_f32;
  if (splitk <= 0) splitk = dlia_gemm_skinny_splitk(M, N, K);
  if (workspace == nullptr) splitk = 1;
  const int chunks = K / kSkinnyBK;
  if (splitk > chunks) splitk = chunks;
  p.splitk = splitk;
  if (M <= 16)      launch_skinny<1, 2>(p, stream);
  else if (M <= 32) launch_skinny<2, 2>(p, stream);
  else              launch_skinny<4, 2>(p, stream);
  if (splitk > 1) {
    const int nout = p.act == 1 ? N / 2 : N;
    const long work = (long)M * (nout / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, p);
  }
}
