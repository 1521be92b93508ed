// Parameter block and fused epilogue shared by the weight-streaming (decode) and tiled
// (prefill) GEMM kernels. Both produce, per lane, 4 consecutive output columns of one row.
#pragma once
#include "common.h"

namespace dlia {

struct GemmParams {
  const bf16* x; long ldx;
  const u32x4* wp;            // packed weights (bf16 fragments, or fp8 e4m3 with two k-steps per 16 B)
  const float* wscale;        // fp8 weights: per packed row (GEMM column) dequantisation scale; else null
  const float* xscale;        // fp8 activations: per row dequantisation scale (x then points at e4m3 bytes)
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
  int group_n;                // gemm_rows.hip only: column blocks per workgroup-order group (set by its launcher)
  int* counters;              // gemm_pingpong.hip with splitk > 1: one arrival counter per output tile, zero between launches
  // gemm_pingpong.hip as the lm_head: besides the fp32 logits, the maximum of every 32-column chunk of every
  // row (columns >= n_valid excluded) -> chunk_max[m * ldc + n / 32]; the sampler then reads only the chunks
  // that can hold a top-k candidate instead of the whole row. Null: not wanted.
  float* chunk_max = nullptr; long ldc = 0; int n_valid = 0;
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
// `res`: the residual's four values when the caller has loaded them already (null: loaded here).
__device__ __forceinline__ void epilogue4(const GemmParams& p, int m, int n, float (&g)[4], const float (&u)[4],
                                          const u32x2* res = nullptr) {
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
    const u32x2 b = res ? *res : *reinterpret_cast<const u32x2*>(p.residual + (long)m * p.ldr + n);
    g[0] += bf16_lo(b[0]); g[1] += bf16_hi(b[0]); g[2] += bf16_lo(b[1]); g[3] += bf16_hi(b[1]);
  }
  store4(p.out, p.ldo, m, n, g, p.out_f32);
}


// Output stage shared by the streaming kernels: acc[i][j][r] = C[m = m0 + i*16 + l15][n = (tile0+j)*16 + quad*4 + r].
// Writes fp32 partials (split-K or partial-only), or applies the fused epilogue.
// DIRECT: the accumulators are already complete (the kernel summed its K splits itself).
template <int MT, int NTW, bool DIRECT = false>
__device__ __forceinline__ void write_acc_tiles(const GemmParams& p, f32x4 (&acc)[MT][NTW], int tile0, int l15,
                                                int quad, int ks, int m0 = 0) {
  if (!DIRECT && (p.partial_only || p.splitk > 1)) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + i * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int n = (tile0 + j) * 16 + quad * 4;
        if (n < p.N) *reinterpret_cast<f32x4*>(p.partial + ((long)ks * p.M + m) * p.N + n) = acc[i][j];
      }
    }
    return;
  }
  // The residual of every tile of the call first (the tiled kernels' O and down projections): one load per tile
  // issued and waited for in turn left a wave idle for a memory latency per tile.
  u32x2 rres[MT][NTW];
  if (DIRECT && p.residual && p.act != 1) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int m = min(m0 + i * 16 + l15, p.M - 1), n = min((tile0 + j) * 16 + quad * 4, p.N - 4);
        rres[i][j] = *reinterpret_cast<const u32x2*>(p.residual + (long)m * p.ldr + n);
      }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + i * 16 + l15;
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
          epilogue4(p, m, n, g, g, DIRECT ? &rres[i][j] : nullptr);
        }
      }
    }
  }
}

}  // namespace dlia
extern "C" void dlia_splitk_reduce(const dlia::GemmParams* p, hipStream_t stream);
extern "C" void dlia_gemm_rows(const dlia::GemmParams* p, hipStream_t stream);
extern "C" void dlia_gemm_rows_f8(const dlia::GemmParams* p, hipStream_t stream);
extern "C" int dlia_gemm_rows_set_narrow(int on);
extern "C" void dlia_gemm_pingpong(const dlia::GemmParams* p, hipStream_t stream);
extern "C" int dlia_gemm_large_choice(int which);
extern "C" int dlia_gemm_pingpong_splitk(int M, int N, int K);
extern "C" long dlia_gemm_pingpong_ws_floats(int M, int N, int K);
extern "C" void dlia_gemm_skinny_config(int M, int N, int K, int epi, int allow_split, int* ntw_out, int* splitk_out);
namespace dlia {

// Packing pads N to a multiple of 128 columns (8 n-tiles), so any (waves x tiles) tiling divides it.
static inline int packed_ntiles(int N) { return ((N + 127) / 128) * 8; }

}  // namespace dlia
