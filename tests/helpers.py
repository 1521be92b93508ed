"""Shared test utilities: tiny HF models holding exactly the engine's random weights."""
import json
import os

import torch

from distributed_llm_inference_amd.models.partition import make_plans
from distributed_llm_inference_amd.models.spec import ModelSpec, tiny_spec
from distributed_llm_inference_amd.models.weights import (export_hf_state_dict, random_stage_weights)
from distributed_llm_inference_amd.models.safetensors_io import save_safetensors
from distributed_llm_inference_amd.runtime.batch import StepBatch


def full_weights(spec, seed=0, dtype=torch.float32):
    plan = make_plans(spec, 1)[0]
    return random_stage_weights(spec, plan, seed=seed, dtype=dtype)


def hf_model_from(spec: ModelSpec, seed=0):
    """A transformers model (fp32, CPU, eval) whose parameters equal `random_stage_weights(seed)`."""
    import transformers
    cfg = spec.hf_config()
    if spec.is_llama:
        conf, cls = {"llama": (transformers.LlamaConfig, transformers.LlamaForCausalLM),
                     "mistral": (transformers.MistralConfig, transformers.MistralForCausalLM),
                     "qwen2": (transformers.Qwen2Config, transformers.Qwen2ForCausalLM)}[spec.hf_type]
        c = conf(**{k: v for k, v in cfg.items() if k not in ("model_type", "architectures")})
        c._attn_implementation = "eager"
        model = cls(c)
    else:
        c = transformers.GPT2Config(**{k: v for k, v in cfg.items() if k not in ("model_type", "architectures")})
        c._attn_implementation = "eager"
        model = transformers.GPT2LMHeadModel(c)
    sd = export_hf_state_dict([full_weights(spec, seed)])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad = [m for m in missing if "rotary" not in m and "lm_head" not in m and "attn.bias" not in m and "masked_bias" not in m]
    assert not bad and not unexpected, (bad, unexpected)
    if spec.tie_embeddings or spec.is_gpt2:
        model.tie_weights()
    return model.eval()


def write_checkpoint(spec: ModelSpec, path: str, seed=0, shards=1, dtype=torch.float32):
    """Write an HF-format checkpoint directory (config.json + safetensors[, index])."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(spec.hf_config(), f)
    sd = {k: v.to(dtype) for k, v in export_hf_state_dict([full_weights(spec, seed)]).items()}
    if shards == 1:
        save_safetensors(sd, os.path.join(path, "model.safetensors"), {"format": "pt"})
    else:
        names = sorted(sd)
        per = (len(names) + shards - 1) // shards
        wm = {}
        for i in range(shards):
            part = names[i * per:(i + 1) * per]
            fn = f"model-{i + 1:05d}-of-{shards:05d}.safetensors"
            save_safetensors({n: sd[n] for n in part}, os.path.join(path, fn), {"format": "pt"})
            wm.update({n: fn for n in part})
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {}, "weight_map": wm}, f)
    return path


def step(slots, q_lens, ctx_lens, token_ids=None, sample=None, temperature=0.0, mb=0, step_id=0,
         tables=None, top_k=50, top_p=0.9, seeds=None, sample_steps=None):
    B = len(slots)
    return StepBatch(step_id=step_id, micro_batch=mb, slots=list(slots), q_lens=list(q_lens), ctx_lens=list(ctx_lens),
                     sample=list(sample) if sample is not None else [1] * B,
                     temperature=[temperature] * B, top_k=[top_k] * B, top_p=[top_p] * B,
                     seeds=list(seeds) if seeds is not None else [0] * B,
                     sample_steps=list(sample_steps) if sample_steps is not None else [0] * B,
                     table_slots=list(tables.keys()) if tables else [],
                     table_blocks=[list(v) for v in tables.values()] if tables else [],
                     token_ids=list(token_ids) if token_ids is not None else None)


def divergences_are_near_ties(ref_stage, prompts, got, want, rel=0.05, abs_tol=0.02):
    """Greedy decoding in bf16 may pick another token than the fp32 reference where two logits are
    nearly tied, after which the sequences differ for good. This checks that EVERY first divergence
    is such a near-tie: the fp32 reference, evaluated on the common prefix, must score the token the
    engine chose within `rel` x max|logit| + `abs_tol` of its own choice. Returns the number of
    diverged sequences; raises AssertionError with the gap otherwise."""
    diverged = 0
    for k, (p, g, w) in enumerate(zip(prompts, got, want)):
        n = 0
        while n < min(len(g), len(w)) and g[n] == w[n]:
            n += 1
        if n == min(len(g), len(w)):
            assert len(g) == len(w), f"sequence {k}: lengths differ without a token mismatch"
            continue
        diverged += 1
        ctx = list(p) + list(w[:n])
        slot = 10_000 + k
        b = step([slot], [len(ctx)], [0])
        logits = ref_stage.forward(b, torch.tensor(ctx, dtype=torch.int32))[0].float().cpu()
        ref_stage.release([slot])
        assert int(logits.argmax()) == w[n], f"sequence {k}: the reference does not reproduce its own token at {n}"
        gap = float(logits[w[n]] - logits[g[n]])
        tol = rel * float(logits.abs().max()) + abs_tol
        assert gap <= tol, (f"sequence {k} diverges at token {n}: engine chose {g[n]}, reference {w[n]}, "
                            f"reference logit gap {gap:.4f} > tolerance {tol:.4f}")
    return diverged


def fallback_rank_main(rank: int, world: int, port: int, tmpdir: str):
    """One rank of the transport-fallback test: the first transport fails on rank 1 only, the second
    on both, the third works; every rank must end up on the third and a failed attempt's links must
    have been aborted and closed. Writes what it saw to tmpdir/rank<r>.json."""
    import json
    import os
    import torch.distributed as dist
    from distributed_llm_inference_amd.parallel import launcher, transport
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = {"closed": 0, "aborted": 0, "jobs": []}

    class FakeLinks(transport.PipelineLinks):
        def abort(self):
            seen["aborted"] += 1

        def close(self):
            seen["closed"] += 1

    def native(r, w, job, device, ctrl_slot_bytes=0, warmup_timeout_s=0):
        seen["jobs"].append(job)
        if r == 1:
            if os.environ.get("DLIA_TEST_FALLBACK_HANG") == "1":      # a communicator creation that never returns
                import time
                time.sleep(3600)
            raise RuntimeError("ncclCommInitRank: unhandled system error (test)")
        return FakeLinks(r, w)

    def torch_groups(r, w, backend, job, host_staged=False, ctrl_slot_bytes=0):
        seen["jobs"].append(job)
        if backend == "nccl":
            raise RuntimeError("Duplicate GPU detected (test)")
        return FakeLinks(r, w)

    transport.make_rccl_links, transport.make_dist_links = native, torch_groups
    links = launcher.rccl_links_or_fallback(rank, world, "job", "cpu", 4096)
    seen["transport"] = links.transport
    with open(os.path.join(tmpdir, f"rank{rank}.json"), "w") as f:
        json.dump(seen, f)
    dist.barrier()
    dist.destroy_process_group()
