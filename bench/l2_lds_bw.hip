// How fast can one CU, and the chip, stream bytes from L2 into LDS? Every workgroup (one per CU:
// it declares 128 KiB of LDS) has W waves; every wave issues 1 KiB pieces with
// global_load_lds_dwordx4 from a region all workgroups share (2 MiB: resident in every XCD's L2;
// 64 MiB: Infinity Cache), keeping at most F pieces in flight (s_waitcnt vmcnt(F - 1) after each
// issue). Prints TB/s chip-wide and bytes per shader cycle and CU (in-kernel clock from
// s_memtime / s_memrealtime). This is the feed rate the 256 x 256 GEMM tile has to live on:
// 64 KiB per 2048 matrix-pipe cycles = 32 B per cycle and CU.
//
//   hipcc --offload-arch=gfx950 -O3 bench/l2_lds_bw.hip -o bench/l2_lds_bw && bench/l2_lds_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_addr), "v"(gptr) : "memory");
}

// P pieces per M0 setting: the instruction's immediate offset advances the global AND the LDS address,
// so consecutive 1 KiB pieces of one source run need no new M0 (does a wave issue faster that way?)
template <int F>
__global__ __launch_bounds__(1024) void stream4_kernel(const char* src, long region, int iters, long* clocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), waves = blockDim.x >> 6;
  const unsigned lds0 = (unsigned)(uintptr_t)smem + wave * (131072 / waves);
  const int slots = 131072 / waves / 4096;
  long off = ((long)blockIdx.x * waves + wave) * 4096 % region;
  const long step = ((long)gridDim.x * waves * 4096) % region;
  const long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 4) {
    const char* g = src + off + lane * 16;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072"
                 :: "s"(lds0 + ((it >> 2) % slots) * 4096), "v"(g) : "memory");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(F >= 4 ? F - 4 : 0) : "memory");
    off += step;
    if (off >= region) off -= region;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    const long w = (long)blockIdx.x * waves + wave;
    clocks[2 * w] = c1 - c0;
    clocks[2 * w + 1] = r1 - r0;
  }
}

template <int F>
__global__ __launch_bounds__(1024) void stream_kernel(const char* src, long region, int iters, long* clocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), waves = blockDim.x >> 6;
  const unsigned lds0 = (unsigned)(uintptr_t)smem + wave * (131072 / waves);
  const int slots = 131072 / waves / 1024;
  // the wave's pieces: a stride of (all waves of the grid) KiB through the region
  long off = ((long)blockIdx.x * waves + wave) * 1024 % region;
  const long step = ((long)gridDim.x * waves * 1024) % region;
  const long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    dma16(src + off + lane * 16, lds0 + (it % slots) * 1024);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(F - 1) : "memory");
    off += step;
    if (off >= region) off -= region;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    const long w = (long)blockIdx.x * waves + wave;
    clocks[2 * w] = c1 - c0;
    clocks[2 * w + 1] = r1 - r0;
  }
}

template <int F, bool FOUR = false>
static void run(const char* src, long region, int waves, int iters, long* dclk, int cus) {
  auto kern = FOUR ? &stream4_kernel<F> : &stream_kernel<F>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<cus, waves * 64, 131072>>>(src, region, iters / 4, dclk);
  hipEventRecord(e0);
  kern<<<cus, waves * 64, 131072>>>(src, region, iters, dclk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long> clk(2L * cus * waves);
  hipMemcpy(clk.data(), dclk, clk.size() * sizeof(long), hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (long w = 0; w < (long)cus * waves; ++w) ghz.push_back(clk[2 * w] / (clk[2 * w + 1] * 10.0));   // ticks per 10 ns
  std::sort(ghz.begin(), ghz.end());
  const double g = ghz[ghz.size() / 2];
  const double bytes = (double)cus * waves * iters * 1024.0;
  printf("%s region %4ld MiB  waves/CU %2d  in flight/wave %2d (%3d KiB/CU): %6.2f TB/s  %5.1f B/cycle/CU at %.2f GHz\n",
         FOUR ? "4 pieces per M0" : "1 piece per M0 ", region >> 20, waves, F, waves * F, bytes / (ms * 1e-3) / 1e12, bytes / cus / (ms * 1e-3) / (g * 1e9), g);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const long cap = 1L << 30;
  char* src; long* dclk;
  hipMalloc(&src, cap + 4096);
  hipMemset(src, 1, cap + 4096);
  hipMalloc(&dclk, 2L * cus * 16 * sizeof(long));
  const int iters = 20000;
  for (int waves : {1, 2, 4, 8}) {
    run<8>(src, 2L << 20, waves, iters, dclk, cus);
    run<8, true>(src, 2L << 20, waves, iters, dclk, cus);
    run<16, true>(src, 2L << 20, waves, iters, dclk, cus);
  }
  for (long region : {2L << 20, 64L << 20, 1L << 30})
    for (int waves : {4, 8, 16}) {
      run<2>(src, region, waves, iters, dclk, cus);
      run<4>(src, region, waves, iters, dclk, cus);
      run<8>(src, region, waves, iters, dclk, cus);
      if (waves * 16 <= 128) run<16>(src, region, waves, iters, dclk, cus);
    }
  return 0;
}
