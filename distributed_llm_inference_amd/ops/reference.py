"""Plain PyTorch fp32 references of every HIP op (used by the numerics tests and by the CPU
stage executor). Shapes follow the kernels: activations are [tokens, features]."""
from __future__ import annotations

import math
import torch
import torch.nn.functional as F


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    return xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w.float()


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def gemm(x, w, bias=None, residual=None, epi: int = 0):
    """epi as in the kernels: 0 none, 1 +bias, 2 +residual, 3 +bias+residual,
    4 swiglu (w = [gate; up] stacked, NOT interleaved), 5 +bias -> gelu_tanh."""
    y = x.float() @ w.float().t()
    if epi in (1, 3, 5):
        y = y + bias.float()
    if epi == 4:
        g, u = y.chunk(2, dim=-1)
        y = F.silu(g) * u
    if epi == 5:
        y = gelu_tanh(y)
    if epi in (2, 3):
        y = y + residual.float()
    return y


def rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
    """x [T, heads, D], half-split convention; cos/sin [P, D/2]."""
    d2 = x.shape[-1] // 2
    c = cos[positions.long()].unsqueeze(1).float()
    s = sin[positions.long()].unsqueeze(1).float()
    xf = x.float()
    a, b = xf[..., :d2], xf[..., d2:]
    return torch.cat([a * c - b * s, b * c + a * s], dim=-1)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_pos0: int, scale: float) -> torch.Tensor:
    """One sequence. q [Tq, Hq, D] are the tokens at absolute positions q_pos0.., k/v
    [Tk, Hkv, D] hold positions 0..Tk-1 (Tk >= q_pos0 + Tq). Causal, GQA. Returns [Tq, Hq, D] fp32."""
    Tq, Hq, D = q.shape
    Tk, Hkv, _ = k.shape
    g = Hq // Hkv
    kf = k.float().repeat_interleave(g, dim=1)
    vf = v.float().repeat_interleave(g, dim=1)
    s = torch.einsum("qhd,khd->hqk", q.float(), kf) * scale
    qpos = torch.arange(Tq, device=q.device).view(-1, 1) + q_pos0
    kpos = torch.arange(Tk, device=q.device).view(1, -1)
    s = s.masked_fill((kpos > qpos).unsqueeze(0), float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("hqk,khd->qhd", p, vf)
