// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of this engine.
// Wavefronts are 64 lanes; every kernel here is written for that width.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dlia {

constexpr int kWave = 64;

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-wide reductions (all 64 lanes end with the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block reductions through LDS; `scratch` must hold >= 16 floats. blockDim.x multiple of 64.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane_id() == 0) scratch[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}

__device__ __forceinline__ float block_max(float v, float* scratch) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane_id() == 0) scratch[w] = v;
  __syncthreads();
  float r = scratch[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
  return r;
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
// streamed-once data (weights in decode): non-temporal so it does not displace reusable lines
__device__ __forceinline__ u32x4 ld16_nt(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}

__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  bf16x2 v = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void unpack8(u32x4 v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_lo(v[i]); f[2 * i + 1] = bf16_hi(v[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
  return v;
}

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// NP consecutive 1 KiB pieces, global -> LDS without a VGPR round trip, under ONE setting of M0: lane l
// of piece q reads 16 bytes at g[q] and lands at lds_addr + 1024 q + 16 l. The pieces are told apart
// by the instruction's 13-bit signed immediate offset, which advances the global AND the LDS address
// (so each source pointer is biased by minus its offset here). Why: a write to M0 waits until every
// earlier LDS-DMA instruction of the wave has read it, ~180 cycles after issue — with M0 set (and
// restored) around every piece a wave issues one piece per ~180 cycles, with one setting per run one
// per ~50 (bench/l2_lds_bw.hip: 32 against 63 bytes per cycle and CU from L2). M0 is clobbered, not
// restored: the kernels that use this have no other user of M0. Every piece counts in vmcnt.
template <int NP>
__device__ __forceinline__ void lds_dma_run(const void* const (&g)[NP], unsigned lds_addr) {
  static_assert(NP == 1 || NP == 2 || NP == 4 || NP == 8, "pieces per run");
  auto at = [&](int q, int off) { return reinterpret_cast<const char*>(g[q]) - off; };
  if constexpr (NP == 8)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off offset:-4096\n\tglobal_load_lds_dwordx4 %2, off offset:-3072\n\t"
                 "global_load_lds_dwordx4 %3, off offset:-2048\n\tglobal_load_lds_dwordx4 %4, off offset:-1024\n\t"
                 "global_load_lds_dwordx4 %5, off\n\tglobal_load_lds_dwordx4 %6, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %7, off offset:2048\n\tglobal_load_lds_dwordx4 %8, off offset:3072"
                 :: "s"(lds_addr + 4096), "v"(at(0, -4096)), "v"(at(1, -3072)), "v"(at(2, -2048)), "v"(at(3, -1024)),
                    "v"(at(4, 0)), "v"(at(5, 1024)), "v"(at(6, 2048)), "v"(at(7, 3072))
                 : "memory", "m0");
  else if constexpr (NP == 4)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, off offset:2048\n\tglobal_load_lds_dwordx4 %4, off offset:3072"
                 :: "s"(lds_addr), "v"(at(0, 0)), "v"(at(1, 1024)), "v"(at(2, 2048)), "v"(at(3, 3072)) : "memory", "m0");
  else if constexpr (NP == 2)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024"
                 :: "s"(lds_addr), "v"(at(0, 0)), "v"(at(1, 1024)) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_addr), "v"(at(0, 0)) : "memory", "m0");
}
// any number of pieces: runs of 8, then 4, 2, 1
template <int NP>
__device__ __forceinline__ void lds_dma_pieces(const void* const (&g)[NP], unsigned lds_addr) {
  if constexpr (NP == 1 || NP == 2 || NP == 4 || NP == 8) {
    lds_dma_run<NP>(g, lds_addr);
  } else {
    static_assert(NP == 16, "1, 2, 4, 8 or 16 pieces");
    const void* lo[8]; const void* hi[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { lo[q] = g[q]; hi[q] = g[8 + q]; }
    lds_dma_run<8>(lo, lds_addr);
    lds_dma_run<8>(hi, lds_addr + 8192);
  }
}

// (v_rcp_f32, 1 ulp, instead of an IEEE division: the dozen instructions a division expands to were a third of the
// vector work of a SwiGLU epilogue, which runs with the matrix cores idle)
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
// GPT-2 "gelu_new": 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
__device__ __forceinline__ float gelu_tanh(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float e = __expf(2.f * u);           // tanh(u) = 1 - 2/(e^{2u}+1)
  const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
  return 0.5f * x * (1.f + t);
}

// Bijective XCD-aware remap of a linear block id: blocks b and b+8 share an XCD (round-robin
// dispatch), so giving each residue class a contiguous chunk of the logical grid keeps
// neighbouring tiles (which share operand panels) in one XCD's L2. Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, i = bid >> 3;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + i;
}

// 4 e4m3 bytes of `d` -> the low (hi = false) or high (hi = true) pair as packed bf16
__device__ __forceinline__ unsigned fp8x2_to_bf16x2(unsigned d, bool hi) {
  const f32x2 f = hi ? __builtin_amdgcn_cvt_pk_f32_fp8(d, true) : __builtin_amdgcn_cvt_pk_f32_fp8(d, false);
  const float a = f.x, b = f.y;
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);   // upper halves
}
__device__ __forceinline__ u32x4 fp8x8_to_bf16x8(unsigned d0, unsigned d1) {
  return u32x4{fp8x2_to_bf16x2(d0, false), fp8x2_to_bf16x2(d0, true), fp8x2_to_bf16x2(d1, false),
               fp8x2_to_bf16x2(d1, true)};
}

}  // namespace dlia
