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

// MODE 0: RMSNorm, 1: LayerNorm (weight + bias), 2: no norm (only fold the addend into x).
// SRC 0: x as is; 1: x += delta (bf16 [rows,H]); 2: x += sum_s partial[s] (+ pbias) where partial
// is the fp32 [S][M][H] split-K output of the preceding GEMM — the reduction of those partials
// costs no launch of its own. In modes 1/2 the updated residual stream is written back into x.
// QOUT: instead of bf16 rows, write the normalised row as OCP e4m3 bytes plus one fp32 scale
// (max|y| / 448) per row — the input format of the fp8 x fp8 GEMM — saving a quantiser launch.
template <int MODE, int SRC, bool QOUT>
__global__ __launch_bounds__(kNormThreads) void norm_kernel(
    bf16* __restrict__ out, bf16* __restrict__ x, const bf16* __restrict__ delta,
    const float* __restrict__ partial, int splits, long pM, const bf16* __restrict__ pbias,
    const bf16* __restrict__ w, const bf16* __restrict__ b, const int* __restrict__ row_idx,
    int H, long ldx, long ldo, float eps, float* __restrict__ qscale) {
  __shared__ float scratch[16];
  const int orow = blockIdx.x;
  const long irow = row_idx ? row_idx[orow] : orow;
  bf16* xr = x + irow * ldx;
  bf16* outr = out + (long)orow * ldo;
  const int nvec = H >> 3;

  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < nvec; i += kNormThreads) {
    float f[8];
    unpack8(ld16(xr + i * 8), f);
    if (SRC == 1) {
      float d[8];
      unpack8(ld16(delta + irow * ldx + i * 8), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += d[j];
    } else if (SRC == 2) {
      const float* pr = partial + irow * H + i * 8;
      const long ss = pM * (long)H;                  // stride between splits
      int s = 0;
      for (; s + 4 <= splits; s += 4) {              // 8 independent 16-B loads in flight
        f32x4 a[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u] = *reinterpret_cast<const f32x4*>(pr + (s + u) * ss);
          c[u] = *reinterpret_cast<const f32x4*>(pr + (s + u) * ss + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f[0] += a[u][0]; f[1] += a[u][1]; f[2] += a[u][2]; f[3] += a[u][3];
          f[4] += c[u][0]; f[5] += c[u][1]; f[6] += c[u][2]; f[7] += c[u][3];
        }
      }
      for (; s < splits; ++s) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(pr + s * ss), c = *reinterpret_cast<const f32x4*>(pr + s * ss + 4);
        f[0] += a[0]; f[1] += a[1]; f[2] += a[2]; f[3] += a[3];
        f[4] += c[0]; f[5] += c[1]; f[6] += c[2]; f[7] += c[3];
      }
      if (pbias) {
        float d[8];
        unpack8(ld16(pbias + i * 8), d);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += d[j];
      }
    }
    if (SRC != 0) {
      u32x4 hv = pack8(f);
      st16(xr + i * 8, hv);           // residual stream updated in place
      unpack8(hv, f);                 // statistics on the rounded value, as an unfused add would see it
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1 += f[j]; s2 += f[j] * f[j]; }
  }
  if (MODE == 2) return;
  float mean = 0.f, rstd;
  if (SRC != 0) __syncthreads();      // the passes below re-read x: make this block's stores visible
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
  auto normed = [&](int i, float (&o)[8]) {
    float f[8], g[8];
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
  };
  if constexpr (QOUT) {
    float amax = 0.f;
    for (int i = threadIdx.x; i < nvec; i += kNormThreads) {
      float o[8];
      normed(i, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(o[j]));
    }
    amax = block_max(amax, scratch);
    const float sc = fmaxf(amax, 1e-12f) * (1.f / 448.f);
    const float inv = 1.f / sc;
    if (threadIdx.x == 0) qscale[orow] = sc;
    unsigned char* q = reinterpret_cast<unsigned char*>(out) + (long)orow * ldo;     // ldo in bytes here
    for (int i = threadIdx.x; i < nvec; i += kNormThreads) {
      float o[8];
      normed(i, o);
      unsigned lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[0] * inv, o[1] * inv, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[2] * inv, o[3] * inv, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[4] * inv, o[5] * inv, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[6] * inv, o[7] * inv, hi, true);
      *reinterpret_cast<u32x2*>(q + i * 8) = u32x2{lo, hi};
    }
  } else {
    for (int i = threadIdx.x; i < nvec; i += kNormThreads) {
      float o[8];
      normed(i, o);
      st16(outr + i * 8, pack8(o));
    }
  }
}

}  // namespace dlia

using namespace dlia;

// mode: 0 RMSNorm, 1 LayerNorm, 2 none (fold the addend into x only). Addend: `delta` (bf16) or
// `partial` (fp32 [splits][pM][H] + optional bias), at most one of them.
extern "C" void dlia_norm(void* out, void* x, const void* delta, const float* partial, int splits, long pM,
                          const void* pbias, const void* w, const void* b, const int* row_idx, int rows, int H,
                          long ldx, long ldo, float eps, int mode, float* qscale, hipStream_t stream) {
  if (rows <= 0) return;
  dim3 grid(rows), block(kNormThreads);
  bf16* o = (bf16*)out; bf16* xx = (bf16*)x;
  const bf16* d = (const bf16*)delta; const bf16* ww = (const bf16*)w; const bf16* bb = (const bf16*)b;
  const bf16* pb = (const bf16*)pbias;
  const int src = partial ? 2 : (delta ? 1 : 0);
#define DLIA_NORM(MODE, SRC)                                                                                      \
  do {                                                                                                            \
    if (qscale) hipLaunchKernelGGL((norm_kernel<MODE, SRC, true>), grid, block, 0, stream, o, xx, d, partial,     \
                                   splits, pM, pb, ww, bb, row_idx, H, ldx, ldo, eps, qscale);                    \
    else hipLaunchKernelGGL((norm_kernel<MODE, SRC, false>), grid, block, 0, stream, o, xx, d, partial,           \
                            splits, pM, pb, ww, bb, row_idx, H, ldx, ldo, eps, qscale);                           \
  } while (0)
  if (mode == 0)      { if (src == 0) DLIA_NORM(0, 0); else if (src == 1) DLIA_NORM(0, 1); else DLIA_NORM(0, 2); }
  else if (mode == 1) { if (src == 0) DLIA_NORM(1, 0); else if (src == 1) DLIA_NORM(1, 1); else DLIA_NORM(1, 2); }
  else                { qscale = nullptr; if (src == 1) DLIA_NORM(2, 1); else if (src == 2) DLIA_NORM(2, 2); }
#undef DLIA_NORM
}
