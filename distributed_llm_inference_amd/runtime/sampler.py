"""Token sampling: temperature -> top-k -> top-p -> categorical draw.

`filter_logits` is a line-by-line behavioural transcription of the reference sampler
(orchestration.py:143-169) and is the oracle the device kernel (`ops/csrc/sampler.hip`) is
tested against:

* logits are divided by `temperature`;
* top-k masks everything strictly below the k-th largest value (ties at the k-th value survive);
* top-p sorts descending, takes softmax + cumulative sum, and removes entries whose
  *previous* cumulative mass already exceeds `top_p` (mask shifted right by one; the first token
  that crosses the threshold stays; the top-1 always stays);
* softmax of the survivors, one multinomial draw.

Deliberate differences: `temperature <= 0` selects greedy argmax (the reference divides by zero
and fails the request), and draws can be seeded per request.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F


@dataclass
class SamplingParams:
    temperature: float = 0.7      # orchestration.py:341 default
    top_k: int = 50               # orchestration.py:353
    top_p: float = 0.9            # orchestration.py:354
    max_new_tokens: int = 20      # orchestration.py:340
    seed: Optional[int] = None
    stop_on_eos: bool = True


def filter_logits(logits: torch.Tensor, temperature: float, top_k: int, top_p: float) -> torch.Tensor:
    """Returns the final probability vector [V] the reference would sample from."""
    z = logits.detach().float().clone()
    z = z / temperature
    if top_k > 0:
        k = min(top_k, z.numel())
        kth = torch.topk(z, k)[0][..., -1, None]
        z[z < kth] = float("-inf")
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(z, descending=True)
        cumulative = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cumulative > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = False
        z[sorted_indices[remove]] = float("-inf")
    return F.softmax(z, dim=-1)


def sample_token(logits: torch.Tensor, temperature: float = 0.7, top_k: int = 50, top_p: float = 0.9,
                 generator: Optional[torch.Generator] = None) -> int:
    """One token id from a 1-D logits vector."""
    if temperature <= 0.0:
        return int(torch.argmax(logits.float()).item())
    probs = filter_logits(logits, temperature, top_k, top_p)
    return int(torch.multinomial(probs, num_samples=1, generator=generator).item())


def top_candidates(logits: torch.Tensor, temperature: float, top_k: int, top_p: float, n: int = 5):
    """(probabilities, ids) of the n most likely tokens after filtering — the reference prints
    these for the first three steps as a debugging aid (orchestration.py:172-178)."""
    if temperature <= 0.0:
        i = int(torch.argmax(logits.float()).item())
        return [1.0], [i]
    probs = filter_logits(logits, temperature, top_k, top_p)
    p, i = torch.topk(probs, min(n, probs.numel()))
    return p.tolist(), i.tolist()
