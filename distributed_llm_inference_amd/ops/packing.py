"""Host-side layout transforms that put operands in MFMA-fragment order.

The decode and prefill GEMM kernels read weights in "packed-B" order: for a weight W[N, K]
(``nn.Linear`` layout), block (n-tile nt, k-step kt) is 64 lanes x 8 bf16 where lane
l = quad*16 + n holds W[nt*16 + n][kt*32 + quad*8 .. +8]; blocks are stored [nt][kt]. N is padded
with zero rows to a multiple of 128 so every workgroup tiling divides it. Weights are static,
so this is a one-time cost at load and buys fully coalesced 1 KiB-per-instruction streaming
with no LDS staging or swizzle on the hot path.
"""
from __future__ import annotations

import torch

PACK_N = 128      # N granule of the packed image
PACK_K = 128      # K granule required by the kernels (4 MFMA k-steps per LDS stage)
KV_BLOCK = 32     # tokens per KV-cache block


def padded_n(n: int) -> int:
    return (n + PACK_N - 1) // PACK_N * PACK_N


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[N, K] -> flat packed image of padded_n(N)*K elements (same dtype/device)."""
    N, K = w.shape
    if K % PACK_K:
        raise ValueError(f"K={K} must be a multiple of {PACK_K}")
    Np = padded_n(N)
    if Np != N:
        w = torch.cat([w, w.new_zeros(Np - N, K)], dim=0)
    v = w.reshape(Np // 16, 16, K // 32, 4, 8)          # [nt, n, kt, quad, j]
    return v.permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


def unpack_weight(wp: torch.Tensor, N: int, K: int) -> torch.Tensor:
    Np = padded_n(N)
    v = wp.reshape(Np // 16, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4)
    return v.reshape(Np, K)[:N].contiguous()


FP8_MAX = 448.0   # largest finite OCP e4m3 value


def quantize_fp8(w: torch.Tensor):
    """[N, K] -> (e4m3 values as uint8 [N, K], fp32 scale [N]) with w ~= q * scale[:, None].
    Per-output-channel symmetric scaling: the row's absolute maximum maps to the e4m3 maximum."""
    wf = w.float()
    amax = wf.abs().amax(dim=1).clamp_min(1e-12)
    scale = amax / FP8_MAX
    q = (wf / scale[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def dequantize_fp8(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q.view(torch.float8_e4m3fn).float() * scale[:, None]


def pack_weight_fp8(w: torch.Tensor):
    """[N, K] -> (packed uint8 image of padded_n(N)*K bytes, fp32 scales [padded_n(N)]).

    Same lane mapping as `pack_weight`, but a lane's 16 bytes hold TWO k-steps: block
    (n-tile nt, k-pair kp) = 64 lanes x 16 B where lane l = quad*16 + n holds
    W[nt*16 + n][kp*64 + quad*8 .. +8] in bytes 0-7 and W[..][kp*64 + 32 + quad*8 .. +8] in 8-15."""
    N, K = w.shape
    if K % PACK_K:
        raise ValueError(f"K={K} must be a multiple of {PACK_K}")
    q, scale = quantize_fp8(w)
    Np = padded_n(N)
    if Np != N:
        q = torch.cat([q, q.new_zeros(Np - N, K)], dim=0)
        scale = torch.cat([scale, scale.new_ones(Np - N)])
    v = q.reshape(Np // 16, 16, K // 64, 2, 4, 8)        # [nt, n, kp, half, quad, j]
    return v.permute(0, 2, 4, 1, 3, 5).contiguous().reshape(-1), scale.contiguous()


def unpack_weight_fp8(wp: torch.Tensor, scale: torch.Tensor, N: int, K: int) -> torch.Tensor:
    """Dequantised [N, K] fp32 view of a `pack_weight_fp8` image (tests)."""
    Np = padded_n(N)
    v = wp.reshape(Np // 16, K // 64, 4, 16, 2, 8).permute(0, 3, 1, 4, 2, 5).reshape(Np, K)
    return dequantize_fp8(v, scale)[:N].contiguous()


def pack_weight_a8w8(w: torch.Tensor):
    """[N, K] -> (packed e4m3 image, fp32 scales [padded_n(N)]) for the fp8 x fp8 kernel.

    The block-scaled MFMA takes 128 K-elements per instruction: lane l = quad*16 + n holds
    W[nt*16 + n][kc*128 + quad*32 .. +32] as 32 bytes. They are stored as two 16-byte halves in
    separate 1 KiB planes, [n-tile][K/128][half][64 lanes][16 B], so each ring load is contiguous."""
    N, K = w.shape
    if K % PACK_K:
        raise ValueError(f"K={K} must be a multiple of {PACK_K}")
    q, scale = quantize_fp8(w)
    Np = padded_n(N)
    if Np != N:
        q = torch.cat([q, q.new_zeros(Np - N, K)], dim=0)
        scale = torch.cat([scale, scale.new_ones(Np - N)])
    v = q.reshape(Np // 16, 16, K // 128, 4, 2, 16)       # [nt, n, kc, quad, half, b]
    return v.permute(0, 2, 4, 3, 1, 5).contiguous().reshape(-1), scale.contiguous()


def quantize_rows_fp8(x: torch.Tensor):
    """Reference of the device row quantiser: [M, K] -> (e4m3 bytes [M, K], fp32 scale [M])."""
    return quantize_fp8(x)


def interleave_gate_up(w_gate: torch.Tensor, w_up: torch.Tensor) -> torch.Tensor:
    """[I,H],[I,H] -> [2I,H] whose 16-row tiles alternate gate, up: the wave that owns tiles
    (2t, 2t+1) then holds gate and up of the same 16 output columns in identical fragment
    positions and applies silu(g)*u in registers."""
    I, H = w_gate.shape
    if I % 16:
        raise ValueError("intermediate size must be a multiple of 16")
    g = w_gate.reshape(I // 16, 1, 16, H)
    u = w_up.reshape(I // 16, 1, 16, H)
    return torch.cat([g, u], dim=1).reshape(2 * I, H)


# ---- KV-cache layout (mirrors csrc/kv_layout.h) -------------------------------------------

def k_elem_off(kk: torch.Tensor, d: torch.Tensor, D: int) -> torch.Tensor:
    return (kk >> 4) * (16 * D) + ((d >> 3) * 16 + (kk & 15)) * 8 + (d & 7)


def v_elem_off(kk: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    return (((d >> 4) * 4 + ((kk & 15) >> 2)) * 16 + (d & 15)) * 8 + ((kk >> 4) * 4 + (kk & 3))


def kv_block_to_dense(slab_k: torch.Tensor, slab_v: torch.Tensor, D: int):
    """[..., 32*D] slabs -> ([..., 32, D], [..., 32, D]) row-major K and V."""
    kk = torch.arange(KV_BLOCK, device=slab_k.device).view(-1, 1)
    d = torch.arange(D, device=slab_k.device).view(1, -1)
    ko = k_elem_off(kk, d, D).reshape(-1)
    vo = v_elem_off(kk, d).reshape(-1)
    k = slab_k[..., ko].reshape(*slab_k.shape[:-1], KV_BLOCK, D)
    v = slab_v[..., vo].reshape(*slab_v.shape[:-1], KV_BLOCK, D)
    return k, v


def k8_byte_off(kk, d, D: int):
    """Byte offset of K[kk][d] inside an e4m3 K slab (csrc/kv_layout.h): two fragments per 16-byte lane load."""
    kt, s, q = kk >> 4, d >> 5, (d >> 3) & 3
    lane = q * 16 + (kk & 15)
    return ((kt * (D // 64) + (s >> 1)) * 64 + lane) * 16 + (s & 1) * 8 + (d & 7)


def v8_byte_off(kk, d):
    dt = d >> 4
    lane = ((kk & 15) >> 2) * 16 + (d & 15)
    return ((dt >> 1) * 64 + lane) * 16 + (dt & 1) * 8 + ((kk >> 4) * 4 + (kk & 3))


def kv8_block_to_dense(slab_k: torch.Tensor, slab_v: torch.Tensor, scale_k: torch.Tensor, scale_v: torch.Tensor, D: int):
    """e4m3 slabs [..., 32*D] (uint8) + scales [..., 32] -> dequantised fp32 ([..., 32, D], [..., 32, D])."""
    kk = torch.arange(KV_BLOCK, device=slab_k.device).view(-1, 1)
    d = torch.arange(D, device=slab_k.device).view(1, -1)
    ko = k8_byte_off(kk, d, D).reshape(-1)
    vo = v8_byte_off(kk, d).reshape(-1)
    k = slab_k[..., ko].reshape(*slab_k.shape[:-1], KV_BLOCK, D).view(torch.float8_e4m3fn).float()
    v = slab_v[..., vo].reshape(*slab_v.shape[:-1], KV_BLOCK, D).view(torch.float8_e4m3fn).float()
    return k * scale_k[..., None], v * scale_v[..., None]


def dense_to_kv_block(k: torch.Tensor, v: torch.Tensor):
    """Inverse of kv_block_to_dense: [..., 32, D] -> [..., 32*D] slabs."""
    D = k.shape[-1]
    kk = torch.arange(KV_BLOCK, device=k.device).view(-1, 1)
    d = torch.arange(D, device=k.device).view(1, -1)
    ko = k_elem_off(kk, d, D).reshape(-1)
    vo = v_elem_off(kk, d).reshape(-1)
    sk = torch.empty(*k.shape[:-2], KV_BLOCK * D, dtype=k.dtype, device=k.device)
    sv = torch.empty_like(sk)
    sk[..., ko] = k.reshape(*k.shape[:-2], -1)
    sv[..., vo] = v.reshape(*v.shape[:-2], -1)
    return sk, sv
