"""Unit tests that need no GPU: sampler semantics, chat template, tokenizer fallback, config,
partitioner, safetensors per-stage filter, packing round trips, descriptors, native runtime."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from distributed_llm_inference_amd.config import EngineConfig
from distributed_llm_inference_amd.models.partition import (balanced_split, even_split, make_plans,
                                                            make_worker_plans)
from distributed_llm_inference_amd.models.safetensors_io import CheckpointReader, SafetensorsFile, save_safetensors
from distributed_llm_inference_amd.models.spec import BUILTIN_SPECS, ModelSpec, tiny_spec
from distributed_llm_inference_amd.models.weights import (load_stage_weights, random_stage_weights,
                                                          stage_tensor_names)
from distributed_llm_inference_amd.ops import packing
from distributed_llm_inference_amd.runtime.batch import StepBatch
from distributed_llm_inference_amd.runtime.sampler import filter_logits, sample_token
from distributed_llm_inference_amd.runtime.tokenizer import (ByteTokenizer, build_prompt, format_chat_prompt,
                                                             load_tokenizer)

from helpers import write_checkpoint


# ---- sampler: exact reference semantics ------------------------------------------------------
def reference_filter(next_token_logits, temperature, top_k, top_p):
    """Transcription of the reference's sampling block (orchestration.py:147-168), kept separate
    from the engine's own implementation on purpose."""
    next_token_logits = next_token_logits.clone() / temperature
    if top_k > 0:
        indices_to_remove = next_token_logits < torch.topk(next_token_logits, top_k)[0][..., -1, None]
        next_token_logits[indices_to_remove] = float("-inf")
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(next_token_logits, descending=True)
        cumulative_probs = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        sorted_indices_to_remove = cumulative_probs > top_p
        sorted_indices_to_remove[..., 1:] = sorted_indices_to_remove[..., :-1].clone()
        sorted_indices_to_remove[..., 0] = 0
        indices_to_remove = sorted_indices[sorted_indices_to_remove]
        next_token_logits[indices_to_remove] = float("-inf")
    return F.softmax(next_token_logits, dim=-1)


@pytest.mark.parametrize("seed", range(5))
def test_filter_matches_reference(seed):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(32000, generator=g) * 3
    for t, k, p in [(0.7, 50, 0.9), (1.0, 5, 0.5), (0.3, 1, 0.9), (1.5, 200, 0.99), (0.7, 0, 0.9), (0.7, 50, 1.0)]:
        assert torch.equal(filter_logits(z, t, k, p), reference_filter(z, t, k, p))


def test_topk_ties_survive_and_top1_always_kept():
    z = torch.tensor([5.0, 3.0, 3.0, 3.0, 1.0, 0.0])
    p = filter_logits(z, 1.0, 2, 1.0)
    assert (p > 0).sum() == 4                      # the three tied 3.0s all stay with the 5.0
    z = torch.tensor([10.0, 0.0, 0.0, 0.0])
    p = filter_logits(z, 1.0, 4, 0.5)              # top-1 alone already exceeds top_p
    assert p[0] == 1.0 and (p[1:] == 0).all()
    z = torch.tensor([2.0, 1.9, 1.8, -5.0])
    p = filter_logits(z, 1.0, 4, 0.4)              # shift-by-one: the token that crosses 0.4 stays
    assert (p > 0).tolist() == [True, True, False, False]


def test_greedy_at_zero_temperature_and_multinomial_statistics():
    z = torch.randn(1000)
    assert sample_token(z, temperature=0.0) == int(z.argmax())
    assert sample_token(z, temperature=-1.0) == int(z.argmax())
    g = torch.Generator().manual_seed(0)
    want = filter_logits(z, 0.8, 10, 0.95)
    draws = torch.tensor([sample_token(z, 0.8, 10, 0.95, generator=g) for _ in range(3000)])
    freq = torch.bincount(draws, minlength=1000).float() / 3000
    assert (freq[want == 0] == 0).all() and (freq - want).abs().max() < 0.04
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    assert [sample_token(z, generator=g1) for _ in range(8)] == [sample_token(z, generator=g2) for _ in range(8)]


# ---- chat template / tokenizer ---------------------------------------------------------------
def test_chat_template_is_the_reference_string():
    msg = "Hello"
    assert format_chat_prompt(msg) == f"<|system|>\nYou are a helpful assistant.</s>\n<|user|>\n{msg}</s>\n<|assistant|>\n"


def test_byte_tokenizer_roundtrip_and_specials():
    tk = ByteTokenizer(320)
    text = format_chat_prompt("héllo ✓")
    ids = tk.encode(text)
    assert ids[0] == tk.bos_token_id and ids.count(tk.eos_token_id) == 2
    assert tk.decode(ids, skip_special_tokens=True) == text.replace("</s>", "")
    assert tk.decode(ids[1:], skip_special_tokens=False) == text
    assert tk.decode([3 + 65, 300, 3 + 66]) == "AB"          # ids past the byte range carry no text
    assert isinstance(load_tokenizer(None, 32000, eos_token_id=2), ByteTokenizer)
    assert build_prompt(tk, "x", "sys", "none") == "x"
    assert build_prompt(tk, "x", "sys", "auto").startswith("<|system|>\nsys</s>")


def test_hf_tokenizer_is_used_when_files_exist(tmp_path):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2, "hello": 3, "world": 4}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", bos_token="<s>", eos_token="</s>")
    fast.save_pretrained(str(tmp_path))
    t = load_tokenizer(str(tmp_path), 5)
    assert t.encode("hello world", add_special_tokens=False) == [3, 4]
    assert t.eos_token_id == 2 and t.pad_token_id == 2       # pad <- eos fallback (orchestration.py:35-36)
    assert t.decode([3, 2, 4], skip_special_tokens=True) == "hello world"


# ---- config ------------------------------------------------------------------------------------
def test_config_yaml_env_and_validation(tmp_path, monkeypatch):
    p = tmp_path / "c.yaml"
    p.write_text("model: llama-3-8b\npp_size: 4\nlayer_ranges: [[0, 8], [8, 16], [16, 24], [24, 32]]\n")
    c = EngineConfig.from_yaml(str(p))
    assert c.pp_size == 4 and c.num_micro_batches == 4 and c.partition == "explicit"
    assert c.layer_ranges == [(0, 8), (8, 16), (16, 24), (24, 32)]
    monkeypatch.setenv("DLIA_PP_SIZE", "2")
    monkeypatch.setenv("DLIA_USE_GRAPHS", "false")
    monkeypatch.setenv("DLIA_WORKER_URLS", "http://a:1,http://b:2")
    c2 = EngineConfig.from_env(EngineConfig())
    assert c2.pp_size == 2 and c2.use_graphs is False and c2.worker_urls == ["http://a:1", "http://b:2"]
    assert (c2.temperature, c2.top_k, c2.top_p, c2.default_max_tokens, c2.port) == (0.7, 50, 0.9, 20, 5000)
    with pytest.raises(ValueError):
        EngineConfig.from_dict({"no_such_key": 1})
    with pytest.raises(ValueError):
        EngineConfig(pp_size=0)
    assert EngineConfig(pp_size=1).resolved_transport() == "loopback"
    assert EngineConfig(pp_size=2, device="cpu").resolved_transport() == "gloo"
    # KV cache dtype: validated, settable from the environment, and refused by the torch executor
    assert EngineConfig().kv_dtype == "bf16" and EngineConfig(kv_dtype="fp8").to_dict()["kv_dtype"] == "fp8"
    with pytest.raises(ValueError):
        EngineConfig(kv_dtype="int4")
    monkeypatch.setenv("DLIA_KV_DTYPE", "fp8")
    assert EngineConfig.from_env(EngineConfig()).kv_dtype == "fp8"
    monkeypatch.delenv("DLIA_KV_DTYPE")
    from distributed_llm_inference_amd.parallel.launcher import LocalPipeline
    with pytest.raises(ValueError, match="hip backend"):
        LocalPipeline(EngineConfig(model="tiny-llama-test", device="cpu", backend="torch", transport="loopback", kv_dtype="fp8"))


# ---- model specs / partitioner ---------------------------------------------------------------
def test_builtin_shapes():
    s = BUILTIN_SPECS["llama-3-8b"]
    assert (s.num_layers, s.hidden_size, s.num_heads, s.num_kv_heads, s.head_dim, s.intermediate_size, s.vocab_size) == \
        (32, 4096, 32, 8, 128, 14336, 128256)
    assert abs(s.param_count() / 1e9 - 8.03) < 0.02
    t = BUILTIN_SPECS["tinyllama-1.1b"]
    assert (t.num_layers, t.hidden_size, t.num_kv_heads, t.head_dim) == (22, 2048, 4, 64)
    assert abs(BUILTIN_SPECS["llama-3-70b"].param_count() / 1e9 - 70.55) < 0.1
    assert abs(BUILTIN_SPECS["gpt2"].param_count() / 1e6 - 124.4) < 0.2
    assert ModelSpec.from_pretrained("TinyLlama/TinyLlama-1.1B-Chat-v1.0") is t


def test_spec_roundtrip_through_hf_config(tmp_path):
    for fam in ("llama", "gpt2"):
        s = tiny_spec(fam)
        d = tmp_path / fam
        d.mkdir()
        (d / "config.json").write_text(json.dumps(s.hf_config()))
        r = ModelSpec.from_pretrained(str(d))
        for f in ("family", "num_layers", "hidden_size", "num_heads", "num_kv_heads", "head_dim",
                  "intermediate_size", "vocab_size", "max_position", "tie_embeddings"):
            assert getattr(r, f) == getattr(s, f), f
    with pytest.raises(FileNotFoundError):
        ModelSpec.from_pretrained("no/such-model")


def cost_of(plans, head):
    return [p.num_layers + (head if p.has_head else 0.0) for p in plans]


def test_partitioner():
    assert even_split(22, 2) == [(0, 11), (11, 22)]           # the reference's split (Worker1.py:27-28)
    assert even_split(32, 8) == [(i * 4, i * 4 + 4) for i in range(8)]
    assert even_split(80, 8) == [(i * 10, i * 10 + 10) for i in range(8)]
    assert even_split(7, 3) == [(0, 3), (3, 5), (5, 7)]
    with pytest.raises(ValueError):
        even_split(2, 3)
    s8 = BUILTIN_SPECS["llama-3-8b"]
    b = balanced_split(s8, 4)
    assert b[0][0] == 0 and b[-1][1] == 32 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    from distributed_llm_inference_amd.models.partition import head_cost_layers, stage_costs, weak_scaling_bound
    head = head_cost_layers(s8)
    assert 2.0 < head < 2.4                                   # 2.4 layers' worth of parameters, 2.2 layers of time (bench/stage_cost.py)
    cost = [e - a + (head if i == 3 else 0) for i, (a, e) in enumerate(b)]
    assert max(cost) <= 9.0 + 1e-9 and (b[-1][1] - b[-1][0]) < 8     # the lm_head stage carries fewer layers
    from distributed_llm_inference_amd.models.partition import attn_block_share
    assert attn_block_share(s8) == attn_block_share(s8, 2048) == 0.48 and abs(attn_block_share(s8, 64) - 0.192) < 0.002
    assert attn_block_share(s8, 64) < attn_block_share(s8, 256) < attn_block_share(s8, 512)
    ps = make_plans(s8, 8, split_layers=True, rows=2048)       # the shape bench.py runs on a full node
    assert [p.describe() for p in ps] == ["0-4", "4-8", "8-12", "12-16.5", "16.5-21", "21-25.5", "25.5-30", "30-32"]
    assert weak_scaling_bound(s8, ps, 2048) > 0.94
    p8 = make_plans(s8, 8)
    assert stage_costs(s8, p8) == cost_of(p8, head) and max(stage_costs(s8, p8)) == 5.0
    assert abs(weak_scaling_bound(s8, p8) - (32 + head) / 40) < 1e-9 and weak_scaling_bound(s8, make_plans(s8, 1)) == 1.0
    plans = make_plans(BUILTIN_SPECS["tinyllama"], 2, mode="even")
    assert plans[0].has_embed and not plans[0].has_head and plans[1].has_head and plans[1].describe() == "11-22"
    with pytest.raises(ValueError):
        make_plans(s8, 2, ranges=[(0, 10), (11, 32)])
    edge, workers = make_worker_plans(BUILTIN_SPECS["gpt2"], 2)
    assert [(w.layer_start, w.layer_end) for w in workers] == [(0, 6), (6, 12)]
    assert edge.has_embed and edge.has_head and not any(w.has_embed or w.has_head for w in workers)


# ---- checkpoints --------------------------------------------------------------------------------
@pytest.mark.parametrize("family", ["llama", "gpt2", "qwen2", "llama-attention-bias"])
def test_stage_loader_reads_only_its_tensors(tmp_path, family):
    spec = {"qwen2": lambda: tiny_spec("llama", hf_type="qwen2", qkv_bias=True, tie_embeddings=True),
            "llama-attention-bias": lambda: tiny_spec("llama", qkv_bias=True, o_bias=True)}.get(family, lambda: tiny_spec(family))()
    path = write_checkpoint(spec, str(tmp_path / "ckpt"), seed=2, shards=3)
    plans = make_plans(spec, 2, mode="even")
    full = random_stage_weights(spec, make_plans(spec, 1)[0], seed=2)
    for plan in plans:
        rd = CheckpointReader(path)
        touched = []
        orig = rd.get
        rd.get = lambda n, _o=orig: (touched.append(n), _o(n))[1]
        w = load_stage_weights(path, spec, plan, reader=rd)
        pre = "transformer." if family == "gpt2" else ""
        assert sorted(set(touched)) == sorted(pre + n for n in stage_tensor_names(spec, plan))
        for li, l in zip(plan.layers, w.layers):
            for k, v in l.items():
                assert torch.equal(v, full.layers[li][k]), (li, k)
        assert (w.embed is not None) == (plan.has_embed or (plan.has_head and spec.tie_embeddings))
        assert all(("b_qkv" in l) == (spec.qkv_bias or spec.is_gpt2) and ("b_o" in l) == (spec.o_bias or spec.is_gpt2) for l in w.layers)
        if plan.has_head:
            assert torch.equal(w.lm_head, full.lm_head) and torch.equal(w.final_norm_w, full.final_norm_w)
        rd.close()


def test_end_of_sequence_ids_from_config_and_generation_config(tmp_path):
    import json
    from distributed_llm_inference_amd.models.spec import ModelSpec
    cfg = tiny_spec("llama").hf_config()
    d = tmp_path / "m"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({**cfg, "eos_token_id": [7, 9]}))
    s = ModelSpec.from_pretrained(str(d))
    assert s.eos_token_id == 7 and s.extra_eos_token_ids == (9,)
    assert ModelSpec.from_hf_config(s.hf_config(), name=s.name) == s
    (d / "generation_config.json").write_text(json.dumps({"eos_token_id": [7, 11, 9]}))
    s = ModelSpec.from_pretrained(str(d))
    assert s.eos_token_id == 7 and s.extra_eos_token_ids == (9, 11)
    (d / "config.json").write_text(json.dumps({**cfg, "eos_token_id": None}))
    s = ModelSpec.from_pretrained(str(d))
    assert s.eos_token_id == 7 and s.extra_eos_token_ids == (11, 9)


def test_safetensors_rejects_corrupt_header(tmp_path):
    p = tmp_path / "x.safetensors"
    save_safetensors({"a": torch.arange(6, dtype=torch.float32).reshape(2, 3), "b": torch.ones(4, dtype=torch.bfloat16)}, str(p))
    f = SafetensorsFile(str(p))
    assert torch.equal(f.get("a"), torch.arange(6, dtype=torch.float32).reshape(2, 3)) and f.get("b").dtype == torch.bfloat16
    f.close()
    raw = bytearray(p.read_bytes())
    raw[0:8] = (10 ** 12).to_bytes(8, "little")
    p.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        SafetensorsFile(str(p))


def test_random_weights_are_partition_invariant():
    spec = tiny_spec("llama")
    full = random_stage_weights(spec, make_plans(spec, 1)[0], seed=9)
    for plan in make_plans(spec, 3):
        w = random_stage_weights(spec, plan, seed=9)
        for li, l in zip(plan.layers, w.layers):
            for k in l:
                assert torch.equal(l[k], full.layers[li][k])


# ---- packing / layouts --------------------------------------------------------------------------
def test_weight_packing_roundtrip_and_fragment_order():
    w = torch.arange(200 * 256, dtype=torch.float32).reshape(200, 256)
    wp = packing.pack_weight(w)
    assert wp.numel() == 256 * 256 and torch.equal(packing.unpack_weight(wp, 200, 256), w)
    # block (nt=1, kt=2): lane l=quad*16+n holds W[16+n][64+8*quad : +8]
    KT = 256 // 32
    blk = wp.reshape(-1, 64, 8)[(1 * KT + 2)]
    for lane in (0, 5, 17, 63):
        n, quad = lane & 15, lane >> 4
        assert torch.equal(blk[lane], w[16 + n, 64 + 8 * quad:64 + 8 * quad + 8])
    g, u = torch.randn(64, 128), torch.randn(64, 128)
    iu = packing.interleave_gate_up(g, u)
    assert torch.equal(iu[0:16], g[0:16]) and torch.equal(iu[16:32], u[0:16]) and torch.equal(iu[32:48], g[16:32])


def test_kv_layout_roundtrip():
    for D in (64, 128):
        k, v = torch.randn(3, 32, D), torch.randn(3, 32, D)
        sk, sv = packing.dense_to_kv_block(k, v)
        k2, v2 = packing.kv_block_to_dense(sk, sv, D)
        assert torch.equal(k, k2) and torch.equal(v, v2)
        # K: lane (key, quad) of k-step s reads 8 contiguous elements at s*512 + lane*8
        lane, s, kt = 37, 1, 1
        key, quad = lane & 15, lane >> 4
        off = kt * 16 * D + s * 512 + lane * 8
        assert torch.equal(sk[0, off:off + 8], k[0, kt * 16 + key, s * 32 + quad * 8:s * 32 + quad * 8 + 8])


# ---- descriptors ------------------------------------------------------------------------------
def test_step_descriptor_wire_roundtrip():
    d = StepBatch(step_id=7, micro_batch=3, slots=[4, 9], q_lens=[5, 1], ctx_lens=[0, 77], sample=[1, 0],
                  temperature=[0.7, 0.0], top_k=[50, 1], top_p=[0.9, 1.0], seeds=[2 ** 40 + 3, 5], sample_steps=[0, 9],
                  table_slots=[4], table_blocks=[[10, 11, 12]], token_ids=[1, 2, 3, 4, 5, 6], replay=False)
    d.validate()
    e = StepBatch.from_bytes(d.to_bytes())
    for f in ("kind", "step_id", "micro_batch", "slots", "q_lens", "ctx_lens", "sample", "top_k", "seeds",
              "sample_steps", "table_slots", "table_blocks", "token_ids", "replay"):
        assert getattr(e, f) == getattr(d, f), f
    assert e.temperature == pytest.approx(d.temperature) and e.top_p == pytest.approx(d.top_p)
    d2 = StepBatch(slots=[1], q_lens=[1], ctx_lens=[3], sample=[1], temperature=[0.0], top_k=[1], top_p=[1.0],
                   seeds=[0], sample_steps=[0], token_ids=None, replay=True)
    e2 = StepBatch.from_bytes(d2.to_bytes())
    assert e2.token_ids is None and e2.replay and e2.is_decode
    with pytest.raises(ValueError):
        StepBatch(slots=[1], q_lens=[1, 2]).validate()


def test_compact_replay_descriptor_is_a_header_and_growth_entries():
    """A replay step of a stage that runs its captured graph carries no per-sequence arrays: 40 bytes
    + growth entries instead of 40 bytes per row, whatever the micro-batch size."""
    d = StepBatch.replay_of(step_id=11, micro_batch=2, rows=2048, live=2000)
    d.grow_slots, d.grow_pos, d.grow_blocks = [5, 9], [4, 4], [700, 701]
    d.validate()
    b = d.to_bytes()
    assert len(b) == 40 + 6 * 4
    e = StepBatch.from_bytes(b)
    assert e.compact and e.replay and e.step_id == 11 and e.micro_batch == 2
    assert e.num_seqs == 2048 and e.num_tokens == 2000 and e.num_samples == 2000 and e.is_decode
    assert (e.grow_slots, e.grow_pos, e.grow_blocks) == ([5, 9], [4, 4], [700, 701])
    assert StepBatch.from_bytes(e.to_bytes()).to_bytes() == b           # forwarded down the pipeline unchanged
    with pytest.raises(ValueError):
        StepBatch(replay=False, rows=8, live=8).validate()
    full = StepBatch(slots=list(range(2048)), q_lens=[1] * 2048, ctx_lens=[9] * 2048, sample=[1] * 2048,
                     temperature=[0.7] * 2048, top_k=[50] * 2048, top_p=[0.9] * 2048, seeds=[1] * 2048,
                     sample_steps=[0] * 2048, replay=True)
    assert len(full.to_bytes()) > 80000 and not StepBatch.from_bytes(full.to_bytes()).compact


# ---- native runtime ---------------------------------------------------------------------------
def test_block_allocator_all_or_nothing():
    from distributed_llm_inference_amd.runtime.kv_cache import BlockAllocator, blocks_for
    assert [blocks_for(n) for n in (0, 1, 32, 33)] == [0, 1, 1, 2]
    a = BlockAllocator(8)
    x = a.take(1, 5)
    assert x == [0, 1, 2, 3, 4] and a.take(2, 4) is None and a.free == 3
    y = a.take(2, 3)
    assert a.free == 0 and a.owned(2) == 3
    assert a.release(1) == 5 and a.free == 5 and a.release(1) == 0
    z = a.take(3, 5)
    assert sorted(z) == sorted(x) and a.peak_used == 8 and set(z).isdisjoint(y)


def test_shm_ring_and_status_board():
    from distributed_llm_inference_amd.parallel.shm import ShmRing, StatusBoard
    name = f"dlia_ut_{os.getpid()}"
    prod = ShmRing(name, True, slots=4, slot_bytes=64)
    cons = ShmRing(name, False)
    for i in range(4):
        prod.put(bytes([i]) * (i + 1))
    with pytest.raises(TimeoutError):
        prod.put(b"x", timeout_s=0.05)                 # full
    with pytest.raises(ValueError):
        prod.put(b"y" * 65)
    assert [cons.get(0.1) for _ in range(4)] == [bytes([i]) * (i + 1) for i in range(4)]
    assert cons.get(0.02) is None
    cons.close(); prod.close()
    b = StatusBoard(name + "b", True, 3)
    b.register(1, 11, 22)
    b.set_state(1, 2)
    b.beat(1, 2, 128, 5000)
    r = b.read(1)
    assert (r["state"], r["steps"], r["tokens"], r["layers"], r["pid"]) == ("ready", 2, 128, "11-22", os.getpid())
    assert b.read(2)["state"] == "unregistered"
    b.set_state(1, 3, "boom")
    assert b.read(1)["state"] == "error" and b.read(1)["error"] == "boom"
    b.close()


def test_templates_that_carry_their_own_bos_get_exactly_one():
    """A chat template whose text already starts with the BOS string (Llama-3-Instruct's
    `<|begin_of_text|>`) must not get a second BOS from the tokenizer; the reference's Zephyr
    template has none and gets the tokenizer's usual one."""
    from distributed_llm_inference_amd.runtime.tokenizer import encode_prompt

    class Tok:
        bos_token, bos_token_id = "<|begin_of_text|>", 7

        def encode(self, text, add_special_tokens=True):
            body = [100 + len(text)]
            if text.startswith(self.bos_token):
                body = [self.bos_token_id] + body          # the tokenizer maps the literal to its id
            return ([self.bos_token_id] if add_special_tokens else []) + body

    t = Tok()
    assert encode_prompt(t, "<|begin_of_text|><|start_header_id|>user").count(7) == 1
    assert encode_prompt(t, format_chat_prompt("hi")).count(7) == 1
    b = ByteTokenizer()
    ids = encode_prompt(b, format_chat_prompt("hi"))
    assert ids[0] == b.bos_token_id and ids.count(b.bos_token_id) == 1
    assert encode_prompt(b, "<s>already").count(b.bos_token_id) == 1


def test_balanced_partition_over_attention_and_mlp_blocks():
    """Stage boundaries inside layers: every block belongs to exactly one stage, the bound set by the
    slowest stage improves (Llama-3-8B over 8 stages: 0.86 -> 0.95), and a stage that holds half a
    layer needs only that half's tensors."""
    from distributed_llm_inference_amd.models.partition import (balanced_split_blocks, make_plans, plans_from_blocks,
                                                                weak_scaling_bound)
    from distributed_llm_inference_amd.models.spec import ModelSpec
    from distributed_llm_inference_amd.models.weights import stage_tensor_names
    for name, n in (("llama-3-8b", 8), ("llama-3-8b", 4), ("llama-3-70b", 8), ("tinyllama-1.1b", 2), ("gpt2", 4)):
        spec = ModelSpec.from_pretrained(name)
        whole = make_plans(spec, n)
        halves = make_plans(spec, n, split_layers=True)
        assert weak_scaling_bound(spec, halves) >= weak_scaling_bound(spec, whole) - 1e-9
        attn = [i for p in halves for i in p.attn_layers]
        mlp = [i for p in halves for i in p.mlp_layers]
        assert attn == list(range(spec.num_layers)) and mlp == list(range(spec.num_layers))
        assert halves[0].has_embed and halves[-1].has_head and not halves[0].skip_attn_first and not halves[-1].attn_only_last
        for a, b in zip(halves, halves[1:]):
            assert a.attn_only_last == b.skip_attn_first
            assert b.layer_start == (a.layer_end - 1 if a.attn_only_last else a.layer_end)
        names = [t for p in halves for t in stage_tensor_names(spec, p)]
        assert sorted(set(names)) == sorted(set(stage_tensor_names(spec, make_plans(spec, 1)[0])))
        if not spec.tie_embeddings:
            assert len(names) == len(set(names))            # nothing is loaded twice
    spec = ModelSpec.from_pretrained("llama-3-8b")
    assert weak_scaling_bound(spec, make_plans(spec, 8)) < 0.87 < 0.93 < weak_scaling_bound(spec, make_plans(spec, 8, split_layers=True))
    p = plans_from_blocks(2, [(0, 5), (5, 8)])
    assert [x.describe() for x in p] == ["0-2.5", "2.5-4"] and p[0].attn_layers == [0, 1, 2] and p[0].mlp_layers == [0, 1]
    assert p[1].attn_layers == [3] and p[1].mlp_layers == [2, 3]
    assert balanced_split_blocks(spec, 1) == [(0, 64)]


def test_incremental_detokenizer_never_shows_half_a_character():
    """Streaming text: pieces concatenate to the whole decode, a multi-byte character is held back
    until its last byte is there, and invalid bytes do not hold the stream up for ever."""
    from distributed_llm_inference_amd.runtime.tokenizer import ByteTokenizer, IncrementalDetokenizer
    tok = ByteTokenizer()
    text = "héllo wörld ✓ 日本語 ok"
    ids = tok.encode(text, add_special_tokens=False)
    d = IncrementalDetokenizer(tok)
    pieces = [d.push(ids[:n], final=(n == len(ids))) for n in range(1, len(ids) + 1)]
    assert "".join(pieces) == tok.decode(ids) == text
    assert all("�" not in p for p in pieces) and pieces.count("") == len(ids) - len(text)
    # several tokens at once, and nothing new
    d = IncrementalDetokenizer(tok)
    assert d.push(ids[:7]) + d.push(ids[:7]) + d.push(ids, final=True) == text
    # output that is not text at all still streams, 4 tokens at a time
    bad = [tok.encode("é", add_special_tokens=False)[0]] * 9
    d = IncrementalDetokenizer(tok)
    got = [d.push(bad[:n], final=(n == 9)) for n in range(1, 10)]
    assert "".join(got) == tok.decode(bad) and [len(g) for g in got] == [0, 0, 0, 4, 0, 0, 0, 4, 1]


def test_stop_filter_cuts_at_the_first_stop_string_across_piece_boundaries():
    from distributed_llm_inference_amd.serving.openai_api import StopFilter
    f = StopFilter(["END", "\n\n"])
    out = [f.feed(p) for p in ["Hello wor", "ld. E", "N", "D and more", "ignored"]]
    assert "".join(out) == "Hello world. " and f.hit and f.flush() == ""
    assert all("E" not in o for o in out[1:])            # nothing that might be the start of a stop string goes out early
    f = StopFilter(["END"])
    assert "".join([f.feed("abc E"), f.feed("N"), f.feed("x")]) + f.flush() == "abc ENx" and not f.hit
    f = StopFilter(["b", "abc"])
    assert f.feed("xxabc") == "xx" and f.hit             # the earliest hit wins


def test_chat_messages_use_the_reference_template_turn_by_turn():
    """One user message gives the reference's prompt byte for byte (orchestration.py:66); more turns
    repeat its layout; a system message replaces the default one."""
    from distributed_llm_inference_amd.runtime.tokenizer import ByteTokenizer, build_chat_prompt, format_chat_prompt
    tok = ByteTokenizer()
    default = "You are a helpful assistant."
    assert build_chat_prompt(tok, [{"role": "user", "content": "Hi"}], default) == format_chat_prompt("Hi", default)
    got = build_chat_prompt(tok, [{"role": "system", "content": "Be brief."}, {"role": "user", "content": "a"},
                                  {"role": "assistant", "content": "b"}, {"role": "user", "content": "c"}], default)
    assert got == "<|system|>\nBe brief.</s>\n<|user|>\na</s>\n<|assistant|>\nb</s>\n<|user|>\nc</s>\n<|assistant|>\n"
    assert build_chat_prompt(tok, [{"role": "user", "content": "x"}, {"role": "user", "content": "y"}], default, mode="none") == "x\ny"


def test_hardware_queue_rule(monkeypatch):
    """`ensure_hw_queues`: a rank of a multi-rank job that owns its GPU gets at least 8 hardware queues; a lone
    stage, ranks sharing a card, a larger exported value and DLIA_KEEP_HW_QUEUES=1 keep what is exported."""
    import distributed_llm_inference_amd as pkg

    def after(**env):
        for k in ("GPU_MAX_HW_QUEUES", "WORLD_SIZE", "DLIA_PP_RANKS", "DLIA_SHARE_GPU", "DLIA_KEEP_HW_QUEUES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pkg.ensure_hw_queues()
        import os
        return os.environ.get("GPU_MAX_HW_QUEUES")

    assert after(GPU_MAX_HW_QUEUES="4") == "4"                                   # a lone stage
    assert after(GPU_MAX_HW_QUEUES="4", WORLD_SIZE="1") == "4"
    assert after(GPU_MAX_HW_QUEUES="4", WORLD_SIZE="8") == "8"                   # a torchrun / self-launched rank
    assert after(WORLD_SIZE="2") == "8"                                          # nothing exported
    assert after(GPU_MAX_HW_QUEUES="4", DLIA_PP_RANKS="4") == "8"                # a spawned stage of several
    assert after(GPU_MAX_HW_QUEUES="16", WORLD_SIZE="8") == "16"                 # a larger value stays
    assert after(GPU_MAX_HW_QUEUES="4", WORLD_SIZE="8", DLIA_SHARE_GPU="1") == "4"
    assert after(GPU_MAX_HW_QUEUES="4", WORLD_SIZE="8", DLIA_KEEP_HW_QUEUES="1") == "4"
    assert after(GPU_MAX_HW_QUEUES="x", WORLD_SIZE="8") == "8"                   # an unreadable value counts as unset


def test_a_stage_attaching_to_the_status_board_waits_for_its_header():
    """The shared-memory segment has its size before it has its contents: a stage that attaches in that window
    must wait for the creator's header instead of reading a board of 0 stages (on which `agree_kv_blocks` then
    "agreed" over an empty list — seen once with four ranks coming up together)."""
    import mmap
    import struct
    import threading
    import time
    from distributed_llm_inference_amd.parallel.shm import StatusBoard
    name = f"dlia_test_board_{os.getpid()}"
    path = "/dev/shm/" + name
    with open(path, "wb") as f:
        f.truncate(1 << 20)                      # sized, still all zeros: no magic, no stage count
    got = {}

    def attach():
        b = StatusBoard(name, create=False, timeout_s=10.0)
        got["stages"], got["t"] = b.num_stages, time.time()
        b._owner = False
        b.close()

    try:
        th = threading.Thread(target=attach)
        t0 = time.time()
        th.start()
        time.sleep(0.3)
        assert "stages" not in got               # still waiting
        with open(path, "r+b") as f:
            m = mmap.mmap(f.fileno(), 4096)
            m[4:8] = struct.pack("<I", 3)        # num_stages first, the magic word last
            m[0:4] = struct.pack("<I", 0x424F5244)
            m.flush()
            m.close()
        th.join(10)
        assert got.get("stages") == 3 and got["t"] - t0 >= 0.3
    finally:
        os.unlink(path)
