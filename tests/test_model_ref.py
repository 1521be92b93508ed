"""The torch stage executor against Hugging Face transformers (CPU fp32) on tiny random
checkpoints: full-model logits for prefill and for N cached decode steps. The KV-cache path must
equal full recompute, i.e. what the reference computes by resending the whole sequence
(orchestration.py:109-141)."""
import pytest
import torch

from distributed_llm_inference_amd.models.partition import make_plans
from distributed_llm_inference_amd.models.spec import tiny_spec
from distributed_llm_inference_amd.models.stage_ref import RefStage
from distributed_llm_inference_amd.models.weights import random_stage_weights

from helpers import hf_model_from, step


VARIANTS = {
    "llama": lambda: tiny_spec("llama"),
    "gpt2": lambda: tiny_spec("gpt2"),
    # Llama-family dialects: q/k/v biases (Qwen2), q/k/v and o biases (Llama with attention_bias), Mistral's config
    "qwen2": lambda: tiny_spec("llama", name="tiny-qwen2", hf_type="qwen2", qkv_bias=True, tie_embeddings=True),
    "llama-attention-bias": lambda: tiny_spec("llama", name="tiny-llama-bias", qkv_bias=True, o_bias=True),
    "mistral": lambda: tiny_spec("llama", name="tiny-mistral", hf_type="mistral", sliding_window=64),
}


@pytest.mark.parametrize("family", sorted(VARIANTS))
def test_ref_matches_transformers(family):
    spec = VARIANTS[family]()
    hf = hf_model_from(spec, seed=3)
    plan = make_plans(spec, 1)[0]
    st = RefStage(spec, plan, random_stage_weights(spec, plan, seed=3))
    torch.manual_seed(0)
    ids = torch.randint(3, spec.vocab_size, (1, 19))
    with torch.no_grad():
        want = hf(ids).logits[0]                        # [S, V] full recompute
    # prefill 12 tokens in two chunks (7 + 5), then 7 single-token decode steps
    got = []
    lg = st.forward(step([0], [7], [0], sample=[1]), ids[0, :7])
    got.append((6, lg[0]))
    lg = st.forward(step([0], [5], [7]), ids[0, 7:12])
    got.append((11, lg[0]))
    for p in range(12, 19):
        lg = st.forward(step([0], [1], [p]), ids[0, p:p + 1])
        got.append((p, lg[0]))
    for pos, row in got:
        assert torch.allclose(row, want[pos], atol=2e-4, rtol=1e-4), f"{family} logits differ at position {pos}"


def test_ref_llama3_rope_scaling_matches_transformers():
    from distributed_llm_inference_amd.models.spec import RopeScaling
    spec = tiny_spec("llama", rope_theta=500000.0,
                     rope_scaling=RopeScaling(8.0, 1.0, 4.0, 64), max_position=512)
    hf = hf_model_from(spec, seed=1)
    plan = make_plans(spec, 1)[0]
    st = RefStage(spec, plan, random_stage_weights(spec, plan, seed=1))
    ids = torch.randint(3, spec.vocab_size, (1, 200))
    with torch.no_grad():
        want = hf(ids).logits[0, -1]
    got = st.forward(step([5], [200], [0]), ids[0])[0]
    assert torch.allclose(got, want, atol=3e-4, rtol=1e-4)


@pytest.mark.parametrize("kind", ["linear", "yarn", "yarn-attention-factor"])
def test_ref_linear_and_yarn_rope_scaling_match_transformers(kind):
    from distributed_llm_inference_amd.models.spec import ModelSpec, RopeScaling
    rs = {"linear": RopeScaling(factor=4.0, rope_type="linear", original_max_position_embeddings=512),
          "yarn": RopeScaling(factor=4.0, rope_type="yarn", original_max_position_embeddings=64),
          "yarn-attention-factor": RopeScaling(factor=8.0, rope_type="yarn", original_max_position_embeddings=48,
                                               beta_fast=16.0, beta_slow=2.0, attention_factor=1.3)}[kind]
    spec = tiny_spec("llama", rope_scaling=rs, max_position=512)
    assert ModelSpec.from_hf_config(spec.hf_config(), name=spec.name) == spec
    hf = hf_model_from(spec, seed=1)
    plan = make_plans(spec, 1)[0]
    st = RefStage(spec, plan, random_stage_weights(spec, plan, seed=1))
    ids = torch.randint(3, spec.vocab_size, (1, 200))
    with torch.no_grad():
        want = hf(ids).logits[0, -1]
    got = st.forward(step([5], [200], [0]), ids[0])[0]
    assert torch.allclose(got, want, atol=3e-4, rtol=1e-4)


def test_unsupported_rope_scaling_is_refused():
    from distributed_llm_inference_amd.models.spec import ModelSpec
    cfg = {**tiny_spec("llama").hf_config(), "rope_scaling": {"rope_type": "dynamic", "factor": 2.0}}
    with pytest.raises(ValueError, match="dynamic"):
        ModelSpec.from_hf_config(cfg)
    # architectures that merely look like Llama (same keys) are refused, as are activations the fused MLP does not compute
    base = tiny_spec("llama").hf_config()
    with pytest.raises(ValueError, match="unsupported model_type"):
        ModelSpec.from_hf_config({**base, "model_type": "gemma", "architectures": ["GemmaForCausalLM"]})
    with pytest.raises(ValueError, match="hidden_act"):
        ModelSpec.from_hf_config({**base, "hidden_act": "gelu"})
    assert ModelSpec.from_hf_config({k: v for k, v in base.items() if k not in ("model_type", "architectures")}).is_llama


def test_ref_ragged_batch_equals_single():
    spec = tiny_spec("llama")
    plan = make_plans(spec, 1)[0]
    w = random_stage_weights(spec, plan, seed=0)
    a, b = RefStage(spec, plan, w), RefStage(spec, plan, w)
    x0 = torch.randint(3, spec.vocab_size, (9,))
    x1 = torch.randint(3, spec.vocab_size, (4,))
    la = a.forward(step([0, 1], [9, 4], [0, 0]), torch.cat([x0, x1]))
    l0 = b.forward(step([0], [9], [0]), x0)
    l1 = b.forward(step([1], [4], [0]), x1)
    assert torch.allclose(la[0], l0[0], atol=1e-5) and torch.allclose(la[1], l1[0], atol=1e-5)
    # sample flags select rows
    lc = a.forward(step([0, 1], [1, 1], [9, 4], sample=[0, 1]), torch.tensor([5, 6]))
    assert lc.shape[0] == 1


def test_llama_family_dialects_round_trip_through_config_json():
    """config.json -> spec -> config.json keeps what the engine acts on; a window shorter than the context is refused."""
    from distributed_llm_inference_amd.models.spec import ModelSpec
    for name in ("qwen2", "llama-attention-bias", "mistral", "llama"):
        spec = VARIANTS[name]()
        back = ModelSpec.from_hf_config(spec.hf_config(), name=spec.name)
        assert back == spec, name
    q = ModelSpec.from_hf_config({"model_type": "qwen2", "hidden_size": 128, "num_attention_heads": 4, "num_key_value_heads": 2,
                                  "num_hidden_layers": 2, "intermediate_size": 256, "vocab_size": 320, "sliding_window": 32,
                                  "use_sliding_window": False})
    assert q.qkv_bias and not q.o_bias and q.sliding_window is None
    m = VARIANTS["mistral"]()
    plan = make_plans(m, 1)[0]
    with pytest.raises(ValueError, match="sliding-window"):
        RefStage(m, plan, random_stage_weights(m, plan, seed=0), max_seq_len=128)
    with pytest.raises(ValueError, match="mlp_bias"):
        ModelSpec.from_hf_config({**VARIANTS["llama"]().hf_config(), "mlp_bias": True})
