"""The native RCCL hop (`ops/csrc/comm/rccl_edge.cpp`) on ONE GPU: a one-rank communicator whose
rank exchanges with itself (grouped ncclSend + ncclRecv) runs the real wire path — communicator
creation, the edge's own stream, event fences to and from the compute stream, completion tokens,
capture into a hipGraph — everything except a second device.

Reference behaviour being replaced: orchestration.py:114-137 / Worker1.py:208-245 (JSON over HTTP)."""
import pytest
import torch

from distributed_llm_inference_amd.models.partition import make_plans
from distributed_llm_inference_amd.models.spec import tiny_spec
from distributed_llm_inference_amd.models.weights import random_stage_weights

from helpers import step

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def edge():
    from distributed_llm_inference_amd.parallel import rccl
    e = rccl.RcclEdge(rccl.unique_id(), 1, 0, 0, "cuda:0", name="self")
    yield e
    e.close()


def test_library_binds_the_rccl_pytorch_loaded():
    from distributed_llm_inference_amd.parallel import rccl
    assert rccl.version() >= 21800
    assert len(rccl.unique_id()) == 128


def test_self_hop_round_trips_hidden_states_and_token_ids(edge):
    g = torch.Generator(device="cuda").manual_seed(0)
    src = torch.randn(1024, 4096, device="cuda", generator=g).to(torch.bfloat16)
    dst = torch.zeros_like(src)
    edge.sendrecv(src, dst).sync(60)
    assert torch.equal(src.view(torch.int16), dst.view(torch.int16))
    ids = torch.randint(0, 128256, (1024,), device="cuda", dtype=torch.int32, generator=g)
    out = torch.zeros_like(ids)
    h = edge.sendrecv(ids, out)
    h.wait()                                    # the compute stream waits, not the host
    got = out.clone()
    torch.cuda.synchronize()
    assert torch.equal(got, ids) and h.done()
    st = edge.stats()
    assert st["ops"] >= 2 and st["bytes_sent"] >= src.numel() * 2 + ids.numel() * 4


def test_completion_ring_wraps_without_aliasing_tokens(edge):
    """More transfers than the edge has completion events (256): every transfer still delivers its own
    bytes, a token whose slot has been reused since reads as complete (its transfer finished before the
    slot was handed on) instead of waiting for the newer transfer, and tokens never repeat."""
    src = torch.zeros(64, device="cuda", dtype=torch.int32)
    dst = torch.zeros(700, 64, device="cuda", dtype=torch.int32)
    first = None
    toks = set()
    for i in range(700):
        src.fill_(i)
        h = edge.sendrecv(src, dst[i])
        h.wait()                                 # the next fill_ must not overtake the send
        first = first or h
        toks.add(h.token)
    assert len(toks) == 700
    assert first.done()                          # expired token: complete by construction
    first.wait()
    first.sync(1.0)
    torch.cuda.synchronize()
    assert torch.equal(dst[:, 0].cpu(), torch.arange(700, dtype=torch.int32))
    assert torch.equal(dst.min(dim=1).values, dst.max(dim=1).values)


def test_hop_is_ordered_behind_its_producer_and_ahead_of_its_consumer(edge):
    """The transfer must see what the compute stream wrote just before `send`, and work enqueued
    after `wait()` must see the received bytes — with a long-running producer in front."""
    a = torch.zeros(2048, 4096, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros_like(a)
    big = torch.randn(4096, 4096, device="cuda")
    for i in range(1, 6):
        for _ in range(3):
            big = big @ big * 1e-3              # keep the stream busy so the fence matters
        a.fill_(float(i))
        edge.sendrecv(a, b).wait()
        total = b.float().sum()
        assert float(total.item()) == float(i) * a.numel()


def test_hop_replays_inside_a_captured_graph_with_a_decode_step(edge):
    """recv -> decode step of a last stage -> send, captured once into one hipGraph and replayed:
    hidden states enter through the hop, sampled token ids leave through it. Same tokens as the
    un-captured step on the same inputs."""
    from distributed_llm_inference_amd.models.stage_gpu import GpuStage
    spec = tiny_spec("llama", hidden_size=256, num_heads=4, num_kv_heads=2, head_dim=64,
                     intermediate_size=512, vocab_size=320, num_layers=4)
    plan = make_plans(spec, 2, mode="even")[1]          # last stage: layers 2-4 + head + sampler
    B = 8

    def stage():
        return GpuStage(spec, plan, random_stage_weights(spec, plan, seed=3, dtype=torch.bfloat16), max_slots=B,
                        max_seq_len=128, num_blocks=32, max_tokens=64, max_batch=B, use_graphs=False)

    g = torch.Generator(device="cuda").manual_seed(5)
    hidden = [torch.randn(B, spec.hidden_size, device="cuda", generator=g).to(torch.bfloat16) for _ in range(4)]
    tables = {i: [i * 4, i * 4 + 1, i * 4 + 2, i * 4 + 3] for i in range(B)}

    # plain: eager decode steps fed directly
    ex = stage()
    want = []
    for t, h in enumerate(hidden):
        d = step(list(range(B)), [1] * B, [t] * B, tables=tables if t == 0 else None)
        want.append(ex.decode_step(d, h).clone())

    # captured: the same steps with both hops inside the graph
    ex = stage()
    d0 = step(list(range(B)), [1] * B, [0] * B, tables=tables)
    src = torch.zeros(B, spec.hidden_size, device="cuda", dtype=torch.bfloat16)
    sink = torch.zeros(B, device="cuda", dtype=torch.int32)
    ex.decode_step(d0, hidden[0])                        # uploads the batch composition, warms every kernel
    st = ex._mb_state(0, B)
    st.ctx.zero_(); st.steps.zero_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                        # first use of the hop outside capture
        edge.sendrecv_on_current_stream(src, st.hidden_in)
        edge.sendrecv_on_current_stream(st.tokens, sink)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        edge.sendrecv_on_current_stream(src, st.hidden_in)
        ex._decode_body(st)
        edge.sendrecv_on_current_stream(st.tokens, sink)
    st.ctx.zero_(); st.steps.zero_()
    got = []
    for h in hidden:
        src.copy_(h)
        graph.replay()
        got.append(sink.clone())
    torch.cuda.synchronize()
    for t, (a, b) in enumerate(zip(want, got)):
        assert torch.equal(a, b), f"step {t}: {a.tolist()} vs {b.tolist()}"


def test_abort_then_close_is_clean_and_later_use_fails_loudly():
    from distributed_llm_inference_amd.parallel import rccl
    e = rccl.RcclEdge(rccl.unique_id(), 1, 0, 0, "cuda:0", name="doomed")
    x = torch.ones(64, device="cuda")
    y = torch.zeros(64, device="cuda")
    e.sendrecv(x, y).sync(30)
    e.abort()
    with pytest.raises(rccl.RcclError):
        e.sendrecv(x, y)
    e.close()
    torch.cuda.synchronize()
    assert torch.equal(x, y)
