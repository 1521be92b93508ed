// What the matrix cores of THIS device sustain with nothing else in the way: every wave loops over
// v_mfma_f32_16x16x32_bf16 on operands that stay in registers (16 independent accumulators, two
// waves per SIMD, all 256 CUs), on random and on zero operands, and reports TFLOP/s and the
// in-kernel shader clock (s_memtime ticks per s_memrealtime 100 MHz tick). It is the practical
// ceiling the GEMM kernels in ops/csrc are to be judged against: the chip lowers its clock under
// dense MFMA load, so the 2.5 PFLOP/s figure (2.4 GHz) is not reachable on real data.
//
//   hipcc --offload-arch=gfx950 -O3 bench/mfma_peak.hip -o bench/mfma_peak && bench/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mfma_loop(float* out, long* clocks, int iters, unsigned seed) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      // two bf16 values in [-1, 1) per dword (exponent <= 126), or zeros when seed == 0
      unsigned h = (t * 2654435761u + (i * 4 + j) * 40503u + seed) * 2246822519u;
      const unsigned lo = ((h & 0x807fu) | ((100u + (h >> 16) % 27u) << 7));
      h = h * 3266489917u + 1u;
      const unsigned hi = ((h & 0x807fu) | ((100u + (h >> 16) % 27u) << 7));
      a[i][j] = seed ? (lo | (hi << 16)) : 0u;
      b[i][j] = seed ? (hi | (lo << 16)) : 0u;
    }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                            acc[i][j], 0, 0, 0);
  }
  const long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[t] = s;
  if ((threadIdx.x & 63) == 0) {
    const unsigned w = t >> 6;
    clocks[2 * w] = c1 - c0;
    clocks[2 * w + 1] = r1 - r0;
  }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// the same on v_mfma_f32_32x32x16_bf16: twice the multiply-adds per operand register read
// (4 independent 32 x 32 accumulators per wave)
__global__ __launch_bounds__(512) void mfma32_loop(float* out, long* clocks, int iters, unsigned seed) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 a[2], b[2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) {
      unsigned h = (t * 2654435761u + (i * 4 + j) * 40503u + seed) * 2246822519u;
      const unsigned lo = ((h & 0x807fu) | ((100u + (h >> 16) % 27u) << 7));
      h = h * 3266489917u + 1u;
      const unsigned hi = ((h & 0x807fu) | ((100u + (h >> 16) % 27u) << 7));
      a[i][j] = seed ? (lo | (hi << 16)) : 0u;
      b[i][j] = seed ? (hi | (lo << 16)) : 0u;
    }
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  const long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                              acc[i][j], 0, 0, 0);
  }
  const long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 16; ++k) s += acc[i][j][k];
  out[t] = s;
  if ((threadIdx.x & 63) == 0) {
    const unsigned w = t >> 6;
    clocks[2 * w] = c1 - c0;
    clocks[2 * w + 1] = r1 - r0;
  }
}

typedef int i32x8 __attribute__((ext_vector_type(8)));

// the same loop on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit block scales): 4x the
// multiply-adds per instruction at twice the rate
__global__ __launch_bounds__(512) void mfma8_loop(float* out, long* clocks, int iters, unsigned seed) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  i32x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      unsigned h = (t * 2654435761u + (i * 8 + j) * 40503u + seed) * 2246822519u;
      // four e4m3 bytes, exponent field <= 13 of 15 (no NaN encodings), random sign and mantissa
      unsigned v = 0;
      for (int k = 0; k < 4; ++k) { h = h * 3266489917u + 1u; v |= (((h >> 8) & 0x87u) | ((((h >> 16) % 13u) + 1u) << 3)) << (8 * k); }
      a[i][j] = seed ? (int)v : 0;
      b[i][j] = seed ? (int)((v >> 8) | (v << 24)) : 0;
    }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  const long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[t] = s;
  if ((threadIdx.x & 63) == 0) {
    const unsigned w = t >> 6;
    clocks[2 * w] = c1 - c0;
    clocks[2 * w + 1] = r1 - r0;
  }
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int cus = prop.multiProcessorCount, blocks = cus, threads = 512, iters = 40000;
  const long waves = (long)blocks * threads / 64;
  float* out; long* clocks;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipMalloc(&clocks, waves * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pass = 0; pass < 6; ++pass) {
    const unsigned seed = (pass & 1) == 0 ? 12345u : 0u;
    const bool f8 = pass >= 2 && pass < 4, big = pass >= 4;
    for (int rep = 0; rep < 3; ++rep) {                                    // the last repetition is reported
      hipEventRecord(e0);
      if (big) hipLaunchKernelGGL(mfma32_loop, dim3(blocks), dim3(threads), 0, 0, out, clocks, iters / 2, seed);
      else if (f8) hipLaunchKernelGGL(mfma8_loop, dim3(blocks), dim3(threads), 0, 0, out, clocks, iters / 2, seed);
      else    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, out, clocks, iters, seed);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long> h(2 * waves);
    hipMemcpy(h.data(), clocks, waves * 16, hipMemcpyDeviceToHost);
    double ghz = 0;
    for (long w = 0; w < waves; ++w) ghz += (double)h[2 * w] / (double)h[2 * w + 1] * 0.1;
    ghz /= waves;
    const int n = (f8 || big) ? iters / 2 : iters;
    const int per_iter = big ? 8 : 16;                                       // MFMAs per wave and iteration
    const double flop = (double)waves * n * per_iter * 2.0 * (big ? 32 * 32 * 16 : 16 * 16 * (f8 ? 128 : 32));
    printf("%s %s operands: %d CUs x 8 waves, %.2f ms, %.0f TFLOP/s, in-kernel clock %.2f GHz (%.1f cycles per MFMA and SIMD)\n",
           big ? "bf16 32x32x16 " : f8 ? "e4m3 16x16x128" : "bf16 16x16x32 ", seed ? "random" : "zero  ", cus, ms,
           flop / (ms * 1e-3) / 1e12, ghz, ghz * 1e9 * ms * 1e-3 / ((double)n * per_iter * 2));
  }
  return 0;
}
