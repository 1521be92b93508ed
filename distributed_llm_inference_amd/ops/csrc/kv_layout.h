// Layout of the paged KV cache. One tensor per layer:
//
//   kv[num_blocks][Hkv][2 (K,V)][32 tokens * D]   bf16
//
// Inside a (block, head) slab the 32 tokens are NOT stored row-major: they are stored in the
// exact order in which the 16x16x32 bf16 MFMA wants its operands, so that the attention
// kernel's loads are `base + lane*16 B` — one fully coalesced 1 KiB request per
// wave-instruction, straight into the registers that feed the MFMA, with no LDS staging and no
// shuffles:
//
//  K part = 2 key tiles of 16 keys. Tile kt, MFMA k-step s (32 head-dim elements), lane l
//  (key = l&15, quad = l>>4) holds K[kt*16+key][s*32 + quad*8 .. +8]  (A operand of S^T = K Q^T):
//      elem_off(kk, d) = (kk>>4)*(16*D) + ((d>>3)*16 + (kk&15))*8 + (d&7)
//
//  V part = D/16 tiles of 16 head-dim rows (A operand of O^T = V^T P^T). Lane l (dd = l&15,
//  quad = l>>4) holds, in element j, V[key(quad,j)][dt*16+dd] with key(quad,j) =
//  16*(j>>2) + 4*quad + (j&3) — the order in which the S^T accumulator of two 16-key tiles
//  lands in a lane, so P^T feeds the second MFMA without any data movement:
//      elem_off(kk, d) = (((d>>4)*4 + ((kk&15)>>2))*16 + (d&15))*8 + ((kk>>4)*4 + (kk&3))
//
// fp8 cache (kv_dtype fp8): kv[num_blocks][Hkv][2][32 * D] BYTES of OCP e4m3 plus one fp32 scale per
// (token, kv head, K|V): kv_scale[num_blocks][Hkv][2][32]. Same fragment order at one byte per
// element, with TWO fragments per 16-byte lane load (as the fp8 weights): the K load (key tile kt,
// pair t) holds head-dim k-steps 2t and 2t+1 in bytes 0-7 / 8-15, the V load (pair t) head-dim
// tiles 2t and 2t+1 — a block is 8 wave-loads of 1 KiB instead of 16. The K scale multiplies a row
// of S^T, the V scale is folded into P^T, so both MFMAs run on the exactly widened e4m3 values.
#pragma once

namespace dlia {

constexpr int kBlockTokens = 32;

__host__ __device__ inline long kv_slab_elems(int D) { return (long)kBlockTokens * D; }

__host__ __device__ inline long kv_head_base(long blk, int h, int Hkv, int D, int which) {
  return ((blk * Hkv + h) * 2 + which) * kv_slab_elems(D);
}
__host__ __device__ inline int k_elem_off(int kk, int d, int D) {
  return (kk >> 4) * (16 * D) + ((d >> 3) * 16 + (kk & 15)) * 8 + (d & 7);
}
__host__ __device__ inline int v_elem_off(int kk, int d) {
  return (((d >> 4) * 4 + ((kk & 15) >> 2)) * 16 + (d & 15)) * 8 + ((kk >> 4) * 4 + (kk & 3));
}

__host__ __device__ inline int k8_byte_off(int kk, int d, int D) {
  const int kt = kk >> 4, s = d >> 5, q = (d >> 3) & 3, l = q * 16 + (kk & 15);
  return ((kt * (D / 64) + (s >> 1)) * 64 + l) * 16 + (s & 1) * 8 + (d & 7);
}
__host__ __device__ inline int v8_byte_off(int kk, int d) {
  const int dt = d >> 4, l = ((kk & 15) >> 2) * 16 + (d & 15);
  return ((dt >> 1) * 64 + l) * 16 + (dt & 1) * 8 + ((kk >> 4) * 4 + (kk & 3));
}
__host__ __device__ inline long kv_scale_base(long blk, int h, int Hkv, int which) {
  return ((blk * Hkv + h) * 2 + which) * kBlockTokens;
}

}  // namespace dlia
