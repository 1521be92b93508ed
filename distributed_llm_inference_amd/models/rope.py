"""Rotary position embedding tables.

The reference recomputes cos/sin on every worker for every generated token by calling the HF
rotary module on the whole sequence (Worker1.py:93-117).  Here the table is built once per
stage, `[max_pos, head_dim/2]` in fp32, and the kernels index it by absolute position, so
RoPE never costs a transcendental on the hot path.

Convention: HF "half-split" rotation, `q' = q*cos + rotate_half(q)*sin` with
`rotate_half(x) = cat(-x[d/2:], x[:d/2])`; element i and i+d/2 share frequency i.
"""
from __future__ import annotations

import math
import torch

from .spec import ModelSpec


def inv_frequencies(spec: ModelSpec) -> torch.Tensor:
    d = spec.head_dim
    inv = 1.0 / (spec.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
    rs = spec.rope_scaling
    if rs is not None:
        # Llama-3.1 frequency-dependent rescale: long wavelengths are divided by `factor`,
        # short ones kept, and the band in between is interpolated smoothly.
        low_wavelen = rs.original_max_position_embeddings / rs.low_freq_factor
        high_wavelen = rs.original_max_position_embeddings / rs.high_freq_factor
        wavelen = 2 * math.pi / inv
        scaled = torch.where(wavelen > low_wavelen, inv / rs.factor, inv)
        smooth = (rs.original_max_position_embeddings / wavelen - rs.low_freq_factor) / (
            rs.high_freq_factor - rs.low_freq_factor)
        smoothed = (1 - smooth) * inv / rs.factor + smooth * inv
        mid = (wavelen <= low_wavelen) & (wavelen >= high_wavelen)
        inv = torch.where(mid, smoothed, scaled)
    return inv


def build_rope_table(spec: ModelSpec, max_pos: int, device="cpu"):
    """Returns (cos, sin), each `[max_pos, head_dim/2]` fp32 on `device`."""
    inv = inv_frequencies(spec)
    pos = torch.arange(max_pos, dtype=torch.float64)
    ang = torch.outer(pos, inv)
    return ang.cos().float().to(device), ang.sin().float().to(device)


def apply_rope_ref(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
    """Reference RoPE. x: [T, heads, D]; positions: [T] (absolute)."""
    d2 = x.shape[-1] // 2
    c = cos[positions].unsqueeze(1).to(torch.float32)
    s = sin[positions].unsqueeze(1).to(torch.float32)
    xf = x.to(torch.float32)
    x1, x2 = xf[..., :d2], xf[..., d2:]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1).to(x.dtype)
