// Parameter block and fused epilogue shared by the weight-streaming (decode) and tiled
// (prefill) GEMM kernels. Both produce, per lane, 4 consecutive output columns of one row.
#pragma once
#include "common.h"

namespace dlia {

struct GemmParams {
  const bf16* x; long ldx;
  const u32x4* wp;            // packed weights (bf16 fragments, or fp8 e4m3 with two k-steps per 16 B)
  const float* wscale;        // fp8 weights: per packed row (GEMM column) dequantisation scale; else null
  void* out; long ldo;
  const bf16* bias;           // per output column (pre-activation), may be null
  const bf16* residual; long ldr;
  float* partial;             // [S][M][N] when splitk > 1
  int M, N, K;                // N = number of GEMM columns (for swiglu: 2 * output columns)
  int ntiles;                 // packed n-tiles (N rounded up to the pack granule) / 16
  int splitk;
  int act;                    // 0 none, 1 swiglu (gate tile, up tile interleaved), 2 gelu_tanh
  int out_f32;
  int partial_only;           // write fp32 partials even when splitk == 1 (a consumer kernel sums them)
};

__device__ __forceinline__ void store4(void* out, long ldo, int m, int n, const float (&v)[4], int out_f32) {
  if (out_f32) {
    f32x4 o = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (long)m * ldo + n) = o;
  } else {
    u32x2 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(out) + (long)m * ldo + n) = o;
  }
}

// Shared by the direct epilogue and the split-K reduce kernel: v = 4 consecutive columns
// n..n+3 of row m (for swiglu: g = gate values, u = up values of output columns n..n+3).
__device__ __forceinline__ void epilogue4(const GemmParams& p, int m, int n, float (&g)[4], const float (&u)[4]) {
  if (p.bias) {
    const u32x2 b = *reinterpret_cast<const u32x2*>(p.bias + n);
    g[0] += bf16_lo(b[0]); g[1] += bf16_hi(b[0]); g[2] += bf16_lo(b[1]); g[3] += bf16_hi(b[1]);
  }
  if (p.act == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) g[r] = silu(g[r]) * u[r];
  } else if (p.act == 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) g[r] = gelu_tanh(g[r]);
  }
  if (p.residual) {
    const u32x2 b = *reinterpret_cast<const u32x2*>(p.residual + (long)m * p.ldr + n);
    g[0] += bf16_lo(b[0]); g[1] += bf16_hi(b[0]); g[2] += bf16_lo(b[1]); g[3] += bf16_hi(b[1]);
  }
  store4(p.out, p.ldo, m, n, g, p.out_f32);
}


// Packing pads N to a multiple of 128 columns (8 n-tiles), so any (waves x tiles) tiling divides it.
static inline int packed_ntiles(int N) { return ((N + 127) / 128) * 8; }

}  // namespace dlia
