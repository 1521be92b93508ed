// RMSNorm / LayerNorm (optionally fused with the residual add of the previous sub-block,
// optionally gathering a subset of rows — the last stage normalises only the rows it samples from).
// bf16 in/out, fp32 statistics. One workgroup per row, 16-byte vector accesses.
//
// Replaces the HF LlamaRMSNorm / nn.LayerNorm modules the reference runs on CPU in fp32
// (Worker1.py:128 inside each decoder layer; orchestration.py:140 for the final norm over ALL
// positions).
#include "common.h"
#include "launch.h"

namespace dlia {

constexpr int kNormThreads = 256;

// MODE 0: RMSNorm, MODE 1: LayerNorm (weight + bias)
template <int MODE, bool ADD>
__global__ __launch_bounds__(kNormThreads) void norm_kernel(
    bf16* __restrict__ out, bf16* __restrict__ x, const bf16* __restrict__ delta,
    const bf16* __restrict__ w, const bf16* __restrict__ b, const int* __restrict__ row_idx,
    int H, long ldx, long ldo, float eps) {
  __shared__ float scratch[16];
  const int orow = blockIdx.x;
  const long irow = row_idx ? row_idx[orow] : orow;
  bf16* xr = x + irow * ldx;
  const bf16* dr = ADD ? delta + irow * ldx : nullptr;
  bf16* outr = out + (long)orow * ldo;
  const int nvec = H >> 3;

  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < nvec; i += kNormThreads) {
    float f[8];
    unpack8(ld16(xr + i * 8), f);
    if (ADD) {
      float d[8];
      unpack8(ld16(dr + i * 8), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += d[j];
      u32x4 hv = pack8(f);
      st16(xr + i * 8, hv);           // residual stream updated in place
      unpack8(hv, f);                 // statistics on the rounded value, as an unfused add would see it
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1 += f[j]; s2 += f[j] * f[j]; }
  }
  float mean = 0.f, rstd;
  if (MODE == 0) {
    s2 = block_sum(s2, scratch);
    rstd = rsqrtf(s2 / H + eps);
  } else {
    s1 = block_sum(s1, scratch);
    mean = s1 / H;
    // second pass for the variance: sum((x-mean)^2) is robust where E[x^2]-mean^2 cancels
    float v = 0.f;
    for (int i = threadIdx.x; i < nvec; i += kNormThreads) {
      float f[8];
      unpack8(ld16(xr + i * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float c = f[j] - mean; v += c * c; }
    }
    v = block_sum(v, scratch);
    rstd = rsqrtf(v / H + eps);
  }
  for (int i = threadIdx.x; i < nvec; i += kNormThreads) {
    float f[8], g[8], o[8];
    unpack8(ld16(xr + i * 8), f);
    unpack8(ld16(w + i * 8), g);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f[j] * rstd * g[j];
    } else {
      float bb[8];
      unpack8(ld16(b + i * 8), bb);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (f[j] - mean) * rstd * g[j] + bb[j];
    }
    st16(outr + i * 8, pack8(o));
  }
}

}  // namespace dlia

using namespace dlia;

extern "C" void dlia_norm(void* out, void* x, const void* delta, const void* w, const void* b,
                          const int* row_idx, int rows, int H, long ldx, long ldo, float eps,
                          int layernorm, hipStream_t stream) {
  if (rows <= 0) return;
  dim3 grid(rows), block(kNormThreads);
  bf16* o = (bf16*)out; bf16* xx = (bf16*)x;
  const bf16* d = (const bf16*)delta; const bf16* ww = (const bf16*)w; const bf16* bb = (const bf16*)b;
  if (layernorm) {
    if (delta) hipLaunchKernelGGL((norm_kernel<1, true>), grid, block, 0, stream, o, xx, d, ww, bb, row_idx, H, ldx, ldo, eps);
    else       hipLaunchKernelGGL((norm_kernel<1, false>), grid, block, 0, stream, o, xx, d, ww, bb, row_idx, H, ldx, ldo, eps);
  } else {
    if (delta) hipLaunchKernelGGL((norm_kernel<0, true>), grid, block, 0, stream, o, xx, d, ww, bb, row_idx, H, ldx, ldo, eps);
    else       hipLaunchKernelGGL((norm_kernel<0, false>), grid, block, 0, stream, o, xx, d, ww, bb, row_idx, H, ldx, ldo, eps);
  }
}
