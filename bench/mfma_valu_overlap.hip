// Do MFMA and VALU work from DIFFERENT waves of one SIMD overlap on this device, and from the SAME wave?
//
// A workgroup has 8 waves; wave w and wave w + 4 share SIMD w % 4. Each half ("role") runs one of
//   M  a register-only loop of 16 independent v_mfma_f32_16x16x32_bf16 per iteration (256 MFMA cycles),
//   V  a loop of 32 v_fma_f32 + 8 v_exp_f32 per iteration on 8 independent chains (~256 VALU cycles),
//   MV both bodies in one loop (independent of each other),
//   M32 8 independent-enough v_mfma_f32_32x32x16_bf16 per iteration (the same FLOPs as M), or nothing,
// and reports shader cycles per iteration per role (s_memtime), so a clock that drops under load does
// not blur the comparison. If the pipes overlap across waves, "M | V" costs each role what it costs
// alone; if they serialise, each role sees about the sum. It is the number that decides whether
// issuing the next block's QK^T under the current block's softmax can pay in the attention kernels.
//
//   hipcc --offload-arch=gfx950 -O3 bench/mfma_valu_overlap.hip -o bench/mfma_valu_overlap && bench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { NONE = 0, M = 1, V = 2, MV = 3, M32 = 4 };   // M32: the 32x32x16 form, 8 per iteration (same FLOPs as 16 of M)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WORK>
__device__ __forceinline__ float body(int iters, unsigned t, long& cycles) {
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      unsigned h = (t * 2654435761u + (i * 4 + j) * 40503u + 12345u) * 2246822519u;
      const unsigned lo = ((h & 0x807fu) | ((100u + (h >> 16) % 27u) << 7));
      h = h * 3266489917u + 1u;
      const unsigned hi = ((h & 0x807fu) | ((100u + (h >> 16) % 27u) << 7));
      a[i][j] = lo | (hi << 16);
      b[i][j] = hi | (lo << 16);
    }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 big[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) big[i][j][e] = 0.f;
  float x[8];
  for (int k = 0; k < 8; ++k) x[k] = -1.f - 0.01f * (float)((t + k) & 31);
  const float c = 0.999f + 1e-6f * (float)(t & 3), d = -0.5f;
  const long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (WORK & M) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                              acc[i][j], 0, 0, 0);
    }
    if constexpr (WORK & M32) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            big[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i + 2 * r]), __builtin_bit_cast(bf16x8, b[j + 2 * r]),
                                                                big[i][j], 0, 0, 0);
    }
    if constexpr (WORK & V) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {               // per chain: exp2, then 4 dependent FMAs; values stay in (-2, 0)
        float y = __builtin_amdgcn_exp2f(x[k]);
        y = __builtin_fmaf(y, c, d);
        y = __builtin_fmaf(y, c, d);
        y = __builtin_fmaf(y, c, d);
        x[k] = __builtin_fmaf(y, c, d);
      }
    }
  }
  cycles = __builtin_amdgcn_s_memtime() - c0;
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  for (int k = 0; k < 8; ++k) s += x[k];
  if constexpr (WORK & M32)
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int e = 0; e < 16; ++e) s += big[i][j][e];
  return s;
}

// PRIO1: s_setprio level of waves 4-7 while they work (waves 0-3 stay at 0)
template <int ROLE0, int ROLE1, int PRIO1 = 0>
__global__ __launch_bounds__(512) void overlap_kernel(float* out, long* cycles, int iters) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const int role = (threadIdx.x >> 6) >> 2;       // waves 0-3 / 4-7: one of each per SIMD
  long cyc = 0;
  float s = 0.f;
  if (role == 0) { if constexpr (ROLE0 != NONE) s = body<ROLE0>(iters, t, cyc); }
  else {
    if constexpr (PRIO1 != 0) __builtin_amdgcn_s_setprio(PRIO1);
    if constexpr (ROLE1 != NONE) s = body<ROLE1>(iters, t, cyc);
    if constexpr (PRIO1 != 0) __builtin_amdgcn_s_setprio(0);
  }
  out[t] = s;
  if ((threadIdx.x & 63) == 0) cycles[t >> 6] = cyc;
}

static const char* name(int w) { return w == NONE ? "idle" : w == M ? "MFMA" : w == V ? "VALU" : w == M32 ? "MFMA 32x32x16" : "MFMA+VALU in one loop"; }

template <int R0, int R1, int PRIO1 = 0>
static void run(int blocks, int iters, float* out, long* cyc) {
  const long waves = (long)blocks * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((overlap_kernel<R0, R1, PRIO1>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long> h(waves);
  hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
  double s0 = 0, s1 = 0;
  for (long w = 0; w < waves; ++w) ((w & 7) < 4 ? s0 : s1) += (double)h[w];
  s0 /= (waves / 2.0) * iters; s1 /= (waves / 2.0) * iters;
  printf("waves 0-3: %-22s waves 4-7: %-22s prio %d %8.2f ms   cycles per iteration: %7.1f | %7.1f\n", name(R0), name(R1), PRIO1, ms, s0, s1);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int blocks = prop.multiProcessorCount, iters = 20000;
  float* out; long* cyc;
  hipMalloc(&out, (size_t)blocks * 512 * 4);
  hipMalloc(&cyc, (size_t)blocks * 8 * 8);
  hipMemset(cyc, 0, (size_t)blocks * 8 * 8);
  printf("%d CUs, one workgroup of 8 waves per CU (two waves per SIMD), %d iterations; per iteration 16 MFMA 16x16x32 bf16 and / or 8 v_exp_f32 + 32 v_fma_f32\n", blocks, iters);
  run<M, NONE>(blocks, iters, out, cyc);
  run<NONE, V>(blocks, iters, out, cyc);
  run<M, V>(blocks, iters, out, cyc);
  run<M, M>(blocks, iters, out, cyc);
  run<V, V>(blocks, iters, out, cyc);
  run<V, M>(blocks, iters, out, cyc);             // the VALU wave is the older one
  run<M, V, 3>(blocks, iters, out, cyc);          // the VALU wave has priority
  run<V, M, 3>(blocks, iters, out, cyc);          // the MFMA wave is younger and has priority
  run<M32, NONE>(blocks, iters, out, cyc);
  run<M32, V>(blocks, iters, out, cyc);
  run<M32, V, 3>(blocks, iters, out, cyc);
  run<MV, NONE>(blocks, iters, out, cyc);
  run<MV, MV>(blocks, iters, out, cyc);
  return 0;
}
