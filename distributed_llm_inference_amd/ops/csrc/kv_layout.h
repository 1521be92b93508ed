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

}  // namespace dlia
