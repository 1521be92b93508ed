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
    if rs is not None and rs.rope_type == "linear":
        inv = inv / rs.factor                      # positions divided by the factor
    elif rs is not None and rs.rope_type == "yarn":
        # YaRN: frequencies that turn more than beta_fast times over the original context are kept
        # (extrapolated), those that turn fewer than beta_slow times are divided by the factor
        # (interpolated), a linear ramp over the dimension index in between.
        def corr_dim(rotations):
            return d * math.log(rs.original_max_position_embeddings / (rotations * 2 * math.pi)) / (2 * math.log(spec.rope_theta))
        low, high = corr_dim(rs.beta_fast), corr_dim(rs.beta_slow)
        if rs.truncate:
            low, high = math.floor(low), math.ceil(high)
        low, high = max(low, 0), min(high, d - 1)
        if low == high:
            high += 0.001
        ramp = ((torch.arange(d // 2, dtype=torch.float64) - low) / (high - low)).clamp(0, 1)
        inv = (inv / rs.factor) * ramp + inv * (1 - ramp)
    elif rs is not None:
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


def attention_factor(spec: ModelSpec) -> float:
    """What cos and sin are multiplied by (YaRN: 0.1 ln(factor) + 1 unless the config gives it; 1 otherwise)."""
    rs = spec.rope_scaling
    if rs is None or rs.rope_type != "yarn":
        return 1.0
    if rs.attention_factor is not None:
        return float(rs.attention_factor)
    return 1.0 if rs.factor <= 1 else 0.1 * math.log(rs.factor) + 1.0


def build_rope_table(spec: ModelSpec, max_pos: int, device="cpu"):
    """Returns (cos, sin), each `[max_pos, head_dim/2]` fp32 on `device`."""
    inv = inv_frequencies(spec)
    pos = torch.arange(max_pos, dtype=torch.float64)
    ang = torch.outer(pos, inv)
    f = attention_factor(spec)
    return (ang.cos() * f).float().to(device), (ang.sin() * f).float().to(device)


def apply_rope_ref(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
    """Reference RoPE. x: [T, heads, D]; positions: [T] (absolute)."""
    d2 = x.shape[-1] // 2
    c = cos[positions].unsqueeze(1).to(torch.float32)
    s = sin[positions].unsqueeze(1).to(torch.float32)
    xf = x.to(torch.float32)
    x1, x2 = xf[..., :d2], xf[..., d2:]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1).to(x.dtype)
