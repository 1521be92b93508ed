"""Numerics of every HIP kernel against a plain PyTorch fp32 reference of the same op."""
import math

import pytest
import torch

from distributed_llm_inference_amd import ops
from distributed_llm_inference_amd.ops import packing, reference as ref
from distributed_llm_inference_amd.runtime.sampler import filter_logits

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def K():
    return ops.load()


def bf(x):
    return x.to(torch.bfloat16)


def assert_close(a, b, atol, rtol, what=""):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max().item():.4g} " \
                          f"(ref max {b.abs().max().item():.4g})"


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H", [128, 768, 2048, 4096, 8192])
@pytest.mark.parametrize("mode", ["rms", "layer"])
def test_norm(K, H, mode):
    torch.manual_seed(0)
    T = 37
    x = bf(torch.randn(T, H, device=DEV) * 2 + 0.3)
    w = bf(torch.randn(H, device=DEV) * 0.1 + 1)
    b = bf(torch.randn(H, device=DEV) * 0.1)
    out = torch.empty_like(x)
    K.norm(out, x, None, None, 1, None, w, b if mode == "layer" else None, None, 1e-5, 1 if mode == "layer" else 0)
    want = ref.rmsnorm(x, w, 1e-5) if mode == "rms" else ref.layernorm(x, w, b, 1e-5)
    assert_close(out, want, 2e-2, 1.6e-2, f"norm {mode} H={H}")


def test_norm_fused_add_and_row_gather(K):
    torch.manual_seed(1)
    T, H = 50, 2048
    x = bf(torch.randn(T, H, device=DEV))
    d = bf(torch.randn(T, H, device=DEV))
    w = bf(torch.rand(H, device=DEV) + 0.5)
    x0 = x.clone()
    out = torch.empty_like(x)
    K.norm(out, x, d, None, 1, None, w, None, None, 1e-5, 0)
    h = bf(x0.float() + d.float())
    assert torch.equal(x, h), "residual stream must be updated in place with the rounded sum"
    assert_close(out, ref.rmsnorm(h, w, 1e-5), 2e-2, 1.6e-2, "add_rmsnorm")
    rows = torch.tensor([49, 0, 7], device=DEV, dtype=torch.int32)
    out2 = torch.zeros(3, H, device=DEV, dtype=torch.bfloat16)
    K.norm(out2, x, None, None, 1, None, w, None, rows, 1e-5, 0)
    assert_close(out2, ref.rmsnorm(x[rows.long()], w, 1e-5), 2e-2, 1.6e-2, "row-gather norm")


def test_embed(K):
    torch.manual_seed(2)
    V, H, P, T = 1000, 768, 64, 33
    wte = bf(torch.randn(V, H, device=DEV))
    wpe = bf(torch.randn(P, H, device=DEV))
    ids = torch.randint(0, V, (T,), device=DEV, dtype=torch.int32)
    pos = torch.randint(0, P, (T,), device=DEV, dtype=torch.int32)
    out = torch.empty(T, H, device=DEV, dtype=torch.bfloat16)
    K.embed(out, ids, None, wte, None)
    assert torch.equal(out, wte[ids.long()])
    K.embed(out, ids, pos, wte, wpe)
    assert_close(out, wte[ids.long()].float() + wpe[pos.long()].float(), 1e-2, 1e-2, "embed+wpe")


# ------------------------------------------------------------------------------------------
def _gemm_case(K, M, N, Kd, epi, splitk, out_f32=False, seed=0, ntw=0):
    torch.manual_seed(seed)
    x = bf(torch.randn(M, Kd, device=DEV))
    if epi == 4:
        wg = bf(torch.randn(N // 2, Kd, device=DEV) / math.sqrt(Kd))
        wu = bf(torch.randn(N // 2, Kd, device=DEV) / math.sqrt(Kd))
        wp = packing.pack_weight(packing.interleave_gate_up(wg, wu))
        w_ref = torch.cat([wg, wu], 0)
        n_out = N // 2
    else:
        w = bf(torch.randn(N, Kd, device=DEV) / math.sqrt(Kd))
        wp = packing.pack_weight(w)
        w_ref = w
        n_out = N
    bias = bf(torch.randn(n_out, device=DEV))
    res = bf(torch.randn(M, n_out, device=DEV))
    out = torch.full((M, n_out), float("nan"), device=DEV, dtype=torch.float32 if out_f32 else torch.bfloat16)
    ws = torch.empty(16 * max(M, 128) * N, device=DEV, dtype=torch.float32) if splitk != 1 else None
    K.gemm(out, x, wp, bias, res, ws, N, epi, splitk, ntw)
    want = ref.gemm(x, w_ref, bias, res, epi)
    assert_close(out, want, 3e-2 if not out_f32 else 2e-3, 2e-2 if not out_f32 else 2e-3,
                 f"gemm M={M} N={N} K={Kd} epi={epi} splitk={splitk}")


@pytest.mark.parametrize("M", [1, 7, 16, 33, 64])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4, 5])
def test_gemm_skinny_epilogues(K, M, epi):
    _gemm_case(K, M, 512, 256, epi, 1)


@pytest.mark.parametrize("shape", [(2560, 2048), (2048, 5632), (768, 3072), (4096, 4096)])
@pytest.mark.parametrize("splitk", [1, 0, 4])
def test_gemm_skinny_shapes_splitk(K, shape, splitk):
    N, Kd = shape
    _gemm_case(K, 48, N, Kd, 3, splitk)
    _gemm_case(K, 5, N, Kd, 0, splitk, out_f32=True)


def test_gemm_skinny_swiglu_splitk_and_vocab_pad(K):
    _gemm_case(K, 64, 1024, 1024, 4, 4)
    # N not a multiple of 128: packed image is padded with zero rows, output stays exact
    _gemm_case(K, 3, 1000, 256, 0, 1, out_f32=True)


def _gemm_w8_case(K, M, N, Kd, epi, splitk, out_f32=False, seed=0, ntw=0):
    """fp8 (e4m3, per-channel scale) weights against the fp32 product with the DEQUANTISED weights:
    the kernel widens e4m3 -> bf16 exactly, so only the usual bf16 activation rounding remains."""
    torch.manual_seed(seed)
    x = bf(torch.randn(M, Kd, device=DEV))
    if epi == 4:
        wg = torch.randn(N // 2, Kd, device=DEV) / math.sqrt(Kd)
        wu = torch.randn(N // 2, Kd, device=DEV) / math.sqrt(Kd)
        wp, sc = packing.pack_weight_fp8(packing.interleave_gate_up(wg, wu))
        w_ref = torch.cat([packing.dequantize_fp8(*packing.quantize_fp8(wg)), packing.dequantize_fp8(*packing.quantize_fp8(wu))], 0)
        n_out = N // 2
    else:
        w = torch.randn(N, Kd, device=DEV) / math.sqrt(Kd) * (1 + torch.arange(N, device=DEV).view(-1, 1) % 7)
        wp, sc = packing.pack_weight_fp8(w)
        w_ref = packing.dequantize_fp8(*packing.quantize_fp8(w))
        n_out = N
    bias = bf(torch.randn(n_out, device=DEV))
    res = bf(torch.randn(M, n_out, device=DEV))
    ws = torch.full((16 * max(M, 128) * N,), float("nan"), device=DEV, dtype=torch.float32) if (splitk != 1 or epi == 6) else None
    if epi == 6:
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        K.gemm_w8(out, x, wp, sc, None, None, ws, N, 6, splitk, ntw)
        S = splitk if splitk > 0 else int(K.gemm_config(M, N, Kd, 6, True)[1])
        got = ws[:S * M * N].view(S, M, N).sum(0)
        assert_close(got, x.float() @ w_ref.t(), 2e-3, 2e-3, f"w8 partials M={M} N={N} K={Kd} S={S}")
        return
    out = torch.full((M, n_out), float("nan"), device=DEV, dtype=torch.float32 if out_f32 else torch.bfloat16)
    K.gemm_w8(out, x, wp, sc, bias, res, ws, N, epi, splitk, ntw)
    want = ref.gemm(x, w_ref, bias, res, epi)
    assert_close(out, want, 3e-2 if not out_f32 else 2e-3, 2e-2 if not out_f32 else 2e-3,
                 f"gemm_w8 M={M} N={N} K={Kd} epi={epi} splitk={splitk}")


@pytest.mark.parametrize("M", [1, 16, 33, 64, 128])
@pytest.mark.parametrize("epi", [0, 3, 4, 5, 6])
def test_gemm_w8_epilogues(K, M, epi):
    _gemm_w8_case(K, M, 512, 256, epi, 1, out_f32=(epi == 0))
    _gemm_w8_case(K, M, 2048, 1024, epi, 0 if epi != 6 else 4)


@pytest.mark.parametrize("chunks", [1, 2, 3, 4, 5, 6, 7, 9])
def test_gemm_w8_ring_prefetch_edges(K, chunks):
    for M in (8, 48, 128, 250):
        for ntw in (1, 2):
            _gemm_w8_case(K, M, 2048, 128 * chunks, 0, 1, out_f32=True, ntw=ntw)


def test_gemm_w8_prefill_slices(K):
    """More than 128 rows over fp8 weights: 128-row slices through the streaming kernel."""
    _gemm_w8_case(K, 600, 1024, 512, 3, 1)
    _gemm_w8_case(K, 257, 1024, 512, 4, 1)


def test_fp8_quantisation_error_is_bounded(K):
    """e4m3 with per-row scaling: relative error of a weight is at most 2^-4 of its row maximum
    bucket; the matmul error stays within a few percent of the output scale."""
    torch.manual_seed(1)
    w = torch.randn(512, 1024, device=DEV) / 32
    deq = packing.dequantize_fp8(*packing.quantize_fp8(w))
    rel = ((deq - w).abs() / w.abs().clamp_min(1e-3 * w.abs().max())).max().item()
    assert rel < 0.0726        # half an ulp of a 3-bit mantissa = 2^-4, plus subnormal slack
    x = bf(torch.randn(32, 1024, device=DEV))
    a, b = x.float() @ w.t(), x.float() @ deq.t()
    assert (a - b).abs().max().item() < 0.06 * a.abs().max().item()


def test_quant_rows_matches_reference(K):
    torch.manual_seed(2)
    x = bf(torch.randn(37, 1024, device=DEV) * (1 + torch.arange(37, device=DEV).view(-1, 1)))
    x[5] = 0
    q = torch.zeros(37, 1024, device=DEV, dtype=torch.uint8)
    sc = torch.zeros(37, device=DEV)
    K.quant_rows(q, sc, x)
    q_ref, s_ref = packing.quantize_rows_fp8(x)
    assert_close(sc, s_ref, 1e-6, 1e-5, "row scales")
    deq, deq_ref = packing.dequantize_fp8(q, sc), packing.dequantize_fp8(q_ref, s_ref)
    # device and torch may round an exact tie differently: compare dequantised values within one e4m3 step
    assert ((deq - deq_ref).abs() <= 0.0626 * deq_ref.abs() + 1e-6).all()
    assert (deq[5] == 0).all()


def _gemm_a8w8_case(K, M, N, Kd, epi, splitk, out_f32=False, seed=0, ntw=0):
    """fp8 x fp8 on the block-scaled MFMA against the fp32 product of the DEQUANTISED operands:
    products of e4m3 values are exact in fp32, so only accumulation order differs."""
    torch.manual_seed(seed)
    x = bf(torch.randn(M, Kd, device=DEV) * (1 + torch.arange(M, device=DEV).view(-1, 1) % 5))
    xq = torch.zeros(M, Kd, device=DEV, dtype=torch.uint8)
    xs = torch.zeros(M, device=DEV)
    K.quant_rows(xq, xs, x)
    x_ref = packing.dequantize_fp8(xq, xs)
    if epi == 4:
        wg = torch.randn(N // 2, Kd, device=DEV) / math.sqrt(Kd)
        wu = torch.randn(N // 2, Kd, device=DEV) / math.sqrt(Kd)
        wp, sc = packing.pack_weight_a8w8(packing.interleave_gate_up(wg, wu))
        w_ref = torch.cat([packing.dequantize_fp8(*packing.quantize_fp8(wg)), packing.dequantize_fp8(*packing.quantize_fp8(wu))], 0)
        n_out = N // 2
    else:
        w = torch.randn(N, Kd, device=DEV) / math.sqrt(Kd) * (1 + torch.arange(N, device=DEV).view(-1, 1) % 7)
        wp, sc = packing.pack_weight_a8w8(w)
        w_ref = packing.dequantize_fp8(*packing.quantize_fp8(w))
        n_out = N
    bias = bf(torch.randn(n_out, device=DEV))
    res = bf(torch.randn(M, n_out, device=DEV))
    ws = torch.full((16 * max(M, 128) * N,), float("nan"), device=DEV, dtype=torch.float32) if (splitk != 1 or epi == 6) else None
    if epi == 6:
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        K.gemm_a8w8(out, xq, xs, wp, sc, None, None, ws, N, 6, splitk, ntw)
        S = splitk if splitk > 0 else int(K.gemm_config(M, N, Kd, 6, True)[1])
        got = ws[:S * M * N].view(S, M, N).sum(0)
        assert_close(got, x_ref @ w_ref.t(), 2e-3, 2e-3, f"a8w8 partials M={M} N={N} K={Kd} S={S}")
        return
    out = torch.full((M, n_out), float("nan"), device=DEV, dtype=torch.float32 if out_f32 else torch.bfloat16)
    K.gemm_a8w8(out, xq, xs, wp, sc, bias, res, ws, N, epi, splitk, ntw)
    want = ref.gemm(x_ref, w_ref, bias, res, epi)
    assert_close(out, want, 3e-2 if not out_f32 else 2e-3, 2e-2 if not out_f32 else 2e-3,
                 f"gemm_a8w8 M={M} N={N} K={Kd} epi={epi} splitk={splitk}")


@pytest.mark.parametrize("M", [1, 16, 33, 64, 100, 128, 200, 256])
@pytest.mark.parametrize("epi", [0, 3, 4, 5, 6])
def test_gemm_a8w8_epilogues(K, M, epi):
    _gemm_a8w8_case(K, M, 512, 256, epi, 1, out_f32=(epi == 0))
    _gemm_a8w8_case(K, M, 2048, 1024, epi, 0 if epi != 6 else 4)


@pytest.mark.parametrize("chunks", [1, 2, 3, 4, 5, 6, 7, 9])
def test_gemm_a8w8_ring_prefetch_edges(K, chunks):
    for M in (8, 100, 256):
        for ntw in (1, 2):
            _gemm_a8w8_case(K, M, 2048, 128 * chunks, 0, 1, out_f32=True, ntw=ntw)


def test_gemm_a8w8_layout_identity_and_slices(K):
    """Asymmetric exact-integer operands (every value an exact e4m3 number) pin the operand and
    accumulator maps; 600 rows exercise the 256-row slicing."""
    Kd, N = 256, 256
    x = bf(((torch.arange(64, device=DEV).view(-1, 1) * 3 + torch.arange(Kd, device=DEV).view(1, -1)) % 9).float())
    w = ((torch.arange(N, device=DEV).view(-1, 1) * 5 + torch.arange(Kd, device=DEV).view(1, -1) * 2) % 7).float()
    xq = torch.zeros(64, Kd, device=DEV, dtype=torch.uint8); xs = torch.zeros(64, device=DEV)
    K.quant_rows(xq, xs, x)
    wp, sc = packing.pack_weight_a8w8(w)
    out = torch.empty(64, N, device=DEV, dtype=torch.float32)
    K.gemm_a8w8(out, xq, xs, wp, sc, None, None, None, N, 0, 1, 0)
    want = packing.dequantize_fp8(xq, xs) @ packing.dequantize_fp8(*packing.quantize_fp8(w)).t()
    assert_close(out, want, 1e-3, 1e-4, "a8w8 integer operands")
    _gemm_a8w8_case(K, 600, 1024, 512, 3, 1)


def test_gemm_layout_identity(K):
    """A = I with an ASYMMETRIC B: catches a transposed accumulator map that random-vs-random
    tolerance checks could hide."""
    Kd = N = 256
    x = bf(torch.eye(Kd, device=DEV))[:64]
    w = bf((torch.arange(N, device=DEV).view(-1, 1) * 2 + torch.arange(Kd, device=DEV).view(1, -1) * 3) % 251)
    out = torch.empty(64, N, device=DEV, dtype=torch.float32)
    K.gemm(out, x, packing.pack_weight(w), None, None, None, N, 0, 1, 0)
    assert torch.equal(out, w.float().t()[:64])


@pytest.mark.parametrize("M", [65, 100, 128, 129, 200, 256])
@pytest.mark.parametrize("ntw", [1, 2])
def test_gemm_skinny_128_rows(K, M, ntw):
    _gemm_case(K, M, 1024, 1024, 3, 1, ntw=ntw)
    _gemm_case(K, M, 1024, 1024, 0, 4, out_f32=True, ntw=ntw)
    _gemm_case(K, M, 1024, 512, 4, 2)


@pytest.mark.parametrize("chunks", [1, 2, 3, 4, 5, 6, 7, 9])
def test_gemm_skinny_ring_prefetch_edges(K, chunks):
    """K ranges of 1..9 LDS stages exercise every phase of the 2-deep weight ring (prologue, steady
    pairs, each peeled tail), for every row-tile count and both n-tile widths, on a grid big enough
    (N=2048) that waves really run with loads in flight."""
    for M in (8, 24, 48, 128, 256):
        for ntw in (1, 2):
            _gemm_case(K, M, 2048, 128 * chunks, 0, 1, out_f32=True, ntw=ntw)
    _gemm_case(K, 40, 256, 128 * chunks, 2, 1, ntw=1)
    _gemm_case(K, 48, 2048, 128 * chunks * 2, 0, 2, out_f32=True, ntw=2)


def test_gemm_partials_consumed_by_norm_and_rope(K):
    """epi 6 writes fp32 split-K partials only; the norm and rope kernels sum them."""
    torch.manual_seed(9)
    M, H = 48, 1024
    x = bf(torch.randn(M, H, device=DEV))
    w = bf(torch.randn(H, H, device=DEV) / 32)
    h0 = bf(torch.randn(M, H, device=DEV))
    bias = bf(torch.randn(H, device=DEV))
    g = bf(torch.rand(H, device=DEV) + 0.5)
    for S in (1, 4):
        ws = torch.full((S * M * H,), float("nan"), device=DEV, dtype=torch.float32)
        K.gemm(h0, x, packing.pack_weight(w), None, None, ws, H, 6, S, 0)
        want_sum = ref.gemm(x, w)
        assert_close(ws.view(S, M, H).sum(0), want_sum, 2e-2, 2e-2, f"partials S={S}")
        h = h0.clone()
        out = torch.empty_like(h)
        K.norm(out, h, None, ws, S, bias, g, None, None, 1e-5, 0)
        hn = bf(h0.float() + want_sum + bias.float())
        assert_close(h, hn, 3e-2, 2e-2, "residual folded from partials")
        assert_close(out, ref.rmsnorm(h, g, 1e-5), 2e-2, 1.6e-2, "norm over folded residual")
        h2 = h0.clone()
        K.norm(out, h2, None, ws, S, None, g, None, None, 1e-5, 2)       # fold only
        assert_close(h2, bf(h0.float() + want_sum), 3e-2, 2e-2, "fold-only")
        rows = torch.tensor([5, 47], device=DEV, dtype=torch.int32)
        h3 = h0.clone()
        o3 = torch.zeros(2, H, device=DEV, dtype=torch.bfloat16)
        K.norm(o3, h3, None, ws, S, bias, g, None, rows, 1e-5, 0)
        assert_close(o3, ref.rmsnorm(hn[rows.long()], g, 1e-5), 3e-2, 2e-2, "row-gathered norm over partials")


@pytest.mark.parametrize("M", [257, 300, 520])
@pytest.mark.parametrize("epi", [0, 3, 4, 5])
def test_gemm_a8w8_large_m(K, M, epi):
    """More than 256 rows in one launch (256-row blocks inside the kernel), row counts that end
    inside a block."""
    _gemm_a8w8_case(K, M, 1024, 512, epi, 1)


def test_gemm_a8w8_large_m_llama_shape(K):
    _gemm_a8w8_case(K, 1024, 6144, 4096, 2, 1)
    _gemm_a8w8_case(K, 512, 1152, 1024, 0, 1, out_f32=True)      # 72 n-tiles: 9 column blocks, grid padded to 16


LARGE_KERNELS = {"pingpong": 0, "stream": 2, "quad": 6}     # see the `large_gemm_kernel` op


@pytest.fixture(params=sorted(LARGE_KERNELS))
def large_kernel(K, request):
    """Each kernel for more than 256 rows in turn: the 256 x 256 ping-pong kernel (the default), the
    256 x 128 loader-wave kernel walking 256-row blocks, and the four-wave 256 x 256 kernel (launches that do not split K)."""
    prev = K.large_gemm_kernel(LARGE_KERNELS[request.param])
    yield request.param
    K.large_gemm_kernel(prev)


@pytest.mark.parametrize("M", [257, 300, 520])
@pytest.mark.parametrize("epi", [0, 3, 4, 5])
def test_gemm_large_m(K, large_kernel, M, epi):
    """Row counts that end inside a 256-row (and 128-row) block: the clamped rows are never stored."""
    _gemm_case(K, M, 1024, 512, epi, 1)


def test_gemm_large_m_llama_shape(K, large_kernel):
    _gemm_case(K, 1024, 6144, 4096, 2, 1)


@pytest.mark.parametrize("shape", [(1024, 28672, 4096, 4), (1024, 4096, 14336, 2), (2048, 6144, 4096, 0),
                                   (1000, 1152, 12288, 3), (4100, 1024, 384, 0), (520, 384, 128, 5)])
def test_gemm_pingpong_shapes(K, shape):
    """The default large-M kernel on the Llama-3-8B layer shapes at a 1024-row micro-batch (SwiGLU and
    residual epilogues), a long and a minimal K (two K-tiles), a column count that leaves the last
    256-column block half empty, and row counts that end inside a block."""
    M, N, Kd, epi = shape
    prev = K.large_gemm_kernel(0)
    try:
        _gemm_case(K, M, N, Kd, epi, 1)
    finally:
        K.large_gemm_kernel(prev)


@pytest.mark.parametrize("epi", [0, 2, 4])
@pytest.mark.parametrize("pad", [4, 8])
def test_gemm_large_m_output_rows_that_are_not_16_byte_multiples(K, large_kernel, epi, pad):
    """The row-wise epilogues of the large-M kernels store 16 bytes per lane: an output whose row stride
    is 8 bytes off that (pad 4) must take the general epilogue, one that is a 16-byte multiple but not dense (pad 8)
    the row-wise one with its stride; columns outside the output stay untouched either way."""
    torch.manual_seed(3)
    M, N, Kd = 520, 1024, 512
    n_out = N // 2 if epi == 4 else N
    x = bf(torch.randn(M, Kd, device=DEV))
    if epi == 4:
        wg, wu = (bf(torch.randn(n_out, Kd, device=DEV) / math.sqrt(Kd)) for _ in range(2))
        wp, w_ref = packing.pack_weight(packing.interleave_gate_up(wg, wu)), torch.cat([wg, wu], 0)
    else:
        w_ref = bf(torch.randn(N, Kd, device=DEV) / math.sqrt(Kd))
        wp = packing.pack_weight(w_ref)
    res_big = bf(torch.randn(M, n_out + 8, device=DEV))      # (residual rows must be 16-byte multiples)
    out_big = torch.full((M, n_out + pad), 7.0, device=DEV, dtype=torch.bfloat16)
    res, out = res_big[:, :n_out], out_big[:, :n_out]
    K.gemm(out, x, wp, None, res if epi == 2 else None, None, N, epi, 1, 0)
    want = ref.gemm(x, w_ref, None, res if epi == 2 else None, epi)
    assert_close(out, want, 3e-2, 2e-2, f"strided output epi={epi} pad={pad}")
    assert bool((out_big[:, n_out:] == 7.0).all()), "columns past the output were written"


@pytest.mark.parametrize("shape", [(1024, 4096, 4096, 2), (512, 6144, 4096, 0), (1024, 4096, 14336, 2), (512, 4096, 4096, 3),
                                   (700, 1152, 2048, 4), (300, 384, 1024, 5)])
def test_gemm_pingpong_split_k_inside_the_launch(K, shape):
    """Shapes with fewer 256 x 256 tiles than CUs split K over neighbouring workgroups; the last one
    to finish a tile adds the others' accumulators and runs the fused epilogue. Run repeatedly on one
    workspace: the arrival counters (its last 4096 words) must be back at zero after every launch, and
    an odd number of K-tiles per split (64 tiles / 3) must work."""
    M, N, Kd, epi = shape
    prev = K.large_gemm_kernel(0)
    try:
        need = int(K.gemm_ws_floats(M, N, Kd))
        assert need > 0, "shape does not split: not a test of the split path"
        torch.manual_seed(1)
        x = bf(torch.randn(M, Kd, device=DEV))
        if epi == 4:
            wg, wu = (bf(torch.randn(N // 2, Kd, device=DEV) / math.sqrt(Kd)) for _ in range(2))
            wp, w_ref, n_out = packing.pack_weight(packing.interleave_gate_up(wg, wu)), torch.cat([wg, wu], 0), N // 2
        else:
            w_ref = bf(torch.randn(N, Kd, device=DEV) / math.sqrt(Kd))
            wp, n_out = packing.pack_weight(w_ref), N
        bias, res = bf(torch.randn(n_out, device=DEV)), bf(torch.randn(M, n_out, device=DEV))
        want = ref.gemm(x, w_ref, bias, res, epi)
        ws = torch.zeros(need + 4096, device=DEV, dtype=torch.float32)
        for it in range(4):
            out = torch.full((M, n_out), float("nan"), device=DEV, dtype=torch.bfloat16)
            K.gemm(out, x, wp, bias, res, ws, N, epi, 1, 0)
            assert_close(out, want, 3e-2, 2e-2, f"split-K gemm {shape} launch {it}")
            assert int(ws[-4096:].view(torch.int32).abs().sum()) == 0, "arrival counters not reset"
        # a workspace used by someone else in front (decode partials) is fine: only the tail is sacred
        ws[:need].uniform_(-1e6, 1e6)
        out = torch.full((M, n_out), float("nan"), device=DEV, dtype=torch.bfloat16)
        K.gemm(out, x, wp, bias, res, ws, N, epi, 1, 0)
        assert_close(out, want, 3e-2, 2e-2, f"split-K gemm {shape} dirty workspace")
    finally:
        K.large_gemm_kernel(prev)


@pytest.mark.parametrize("shape", [(1024, 4096, 4096, 2), (512, 6144, 4096, 0), (1024, 4096, 14336, 2), (512, 4096, 4096, 3),
                                   (700, 1152, 4096, 4), (300, 384, 2048, 5)])
def test_gemm_a8w8_pingpong_split_k_inside_the_launch(K, shape):
    """The e4m3 x e4m3 instantiation of the 256 x 256 kernel on shapes that split K inside the launch
    (half as many K-tiles as bf16 for the same K): against the fp32 product of the dequantised
    operands, repeatedly on one workspace, counters back at zero, slabs actually used."""
    M, N, Kd, epi = shape
    prev = K.large_gemm_kernel(0)
    try:
        torch.manual_seed(2)
        x = bf(torch.randn(M, Kd, device=DEV) * (1 + torch.arange(M, device=DEV).view(-1, 1) % 5))
        xq = torch.zeros(M, Kd, device=DEV, dtype=torch.uint8)
        xs = torch.zeros(M, device=DEV)
        K.quant_rows(xq, xs, x)
        x_ref = packing.dequantize_fp8(xq, xs)
        if epi == 4:
            wg, wu = (torch.randn(N // 2, Kd, device=DEV) / math.sqrt(Kd) for _ in range(2))
            wp, sc = packing.pack_weight_a8w8(packing.interleave_gate_up(wg, wu))
            w_ref = torch.cat([packing.dequantize_fp8(*packing.quantize_fp8(wg)), packing.dequantize_fp8(*packing.quantize_fp8(wu))], 0)
            n_out = N // 2
        else:
            w = torch.randn(N, Kd, device=DEV) / math.sqrt(Kd) * (1 + torch.arange(N, device=DEV).view(-1, 1) % 7)
            wp, sc = packing.pack_weight_a8w8(w)
            w_ref, n_out = packing.dequantize_fp8(*packing.quantize_fp8(w)), N
        bias, res = bf(torch.randn(n_out, device=DEV)), bf(torch.randn(M, n_out, device=DEV))
        want = ref.gemm(x_ref, w_ref, bias, res, epi)
        ws = torch.zeros(int(K.gemm_ws_floats(M, N, Kd)) + 4096, device=DEV, dtype=torch.float32)
        for it in range(3):
            out = torch.full((M, n_out), float("nan"), device=DEV, dtype=torch.bfloat16)
            K.gemm_a8w8(out, xq, xs, wp, sc, bias, res, ws, N, epi, 1, 0)
            assert_close(out, want, 3e-2, 2e-2, f"a8w8 split-K gemm {shape} launch {it}")
            assert int(ws[-4096:].view(torch.int32).abs().sum()) == 0, "arrival counters not reset"
        assert float(ws[:-4096].abs().sum()) > 0, "no slab was written: the shape did not split"
        out32 = torch.full((M, n_out), float("nan"), device=DEV, dtype=torch.float32)
        K.gemm_a8w8(out32, xq, xs, wp, sc, bias, res, None, N, epi, 1, 0)          # and unsplit, fp32 out
        assert_close(out32, want, 2e-3, 2e-3, f"a8w8 gemm {shape} without workspace")
    finally:
        K.large_gemm_kernel(prev)


@pytest.mark.parametrize("which", [0, 2], ids=["pingpong", "stream"])
@pytest.mark.parametrize("shape", [(257, 128, 128), (300, 512, 384), (768, 640, 896), (2048, 256, 1024), (513, 1152, 512)])
def test_gemm_a8w8_large_kernels_agree_on_edge_shapes(K, which, shape):
    """One K-tile, an odd number of K-tiles, a column block with one half only (N = 128, 640, 1152),
    row blocks ending mid-tile: both large-M e4m3 kernels against the dequantised product."""
    M, N, Kd = shape
    prev = K.large_gemm_kernel(which)
    try:
        _gemm_a8w8_case(K, M, N, Kd, 3, 1)
        _gemm_a8w8_case(K, M, N, Kd, 0, 1, out_f32=True)
    finally:
        K.large_gemm_kernel(prev)


def test_gemm_pingpong_exact_on_integer_data_with_asymmetric_operands(K):
    """Small integers are exact in bf16 and fp32: every output element must equal the integer product
    exactly, with an x that is not symmetric and weights that differ per column (a swapped row/column
    map, a mis-placed K-tile or a fragment read racing its DMA cannot pass)."""
    from distributed_llm_inference_amd.ops import packing
    M, N, Kd = 768, 512, 896
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randint(-3, 4, (M, Kd), device=DEV, generator=g).float()
    w = torch.randint(-2, 3, (N, Kd), device=DEV, generator=g).float()
    w[:, 0] += torch.arange(N, device=DEV).float() % 5
    want = x @ w.t()
    out = torch.empty(M, N, device=DEV, dtype=torch.float32)
    prev = K.large_gemm_kernel(0)
    try:
        ws = torch.zeros(int(K.gemm_ws_floats(M, N, Kd)) + 4096, device=DEV, dtype=torch.float32)
        for it in range(20):                 # a race would come and go
            out.zero_()
            K.gemm(out, x.to(torch.bfloat16), packing.pack_weight(w.to(torch.bfloat16)), None, None,
                   ws if it % 2 else None, N, 0, 1, 0)      # alternately split over K inside the launch and not
            assert torch.equal(out, want)
    finally:
        K.large_gemm_kernel(prev)


def test_gemm_large_m_row_block_tiers(K):
    """With the narrow policy the streaming kernel walks 256-, 128- or 64-row blocks, whichever is the
    largest that still launches ~200 workgroups. N = 4096 (32 column blocks): 700 and 1024 rows take 128-row blocks
    (700 ends inside one), 400 rows take 64-row blocks, 1500 rows 256-row blocks."""
    prev = K.large_gemm_kernel(4)          # streaming kernel, narrow row-block policy on (off by default)
    try:
        for M in (700, 1024, 400, 1500):
            _gemm_case(K, M, 4096, 512, 2, 1)
        _gemm_case(K, 700, 4096, 512, 4, 1)
        _gemm_case(K, 300, 1024, 512, 3, 1)
        for M in (700, 400, 1500):
            _gemm_a8w8_case(K, M, 4096, 512, 2, 1)
    finally:
        K.large_gemm_kernel(prev)


def test_gemm_large_m_grouped_workgroup_order(K):
    """From 16 row blocks up the streaming kernel orders its workgroups in patches of 4 column blocks
    per XCD. 4096 rows x 4224 columns = 33 column blocks, padded to 40: 5 per XCD, i.e. a full group
    of 4 and a clamped group of 1; every output tile must still be computed exactly once."""
    _gemm_case(K, 4096, 4224, 256, 0, 1)
    _gemm_case(K, 4100, 1024, 256, 2, 1)         # 17 row blocks (the last one 4 rows), 1 column block per XCD
    _gemm_a8w8_case(K, 4096, 4224, 256, 0, 1)


def test_gemm_large_m_column_blocks_not_a_multiple_of_256(K, large_kernel):
    """N = 1152 packs to 72 n-tiles = 4.5 blocks of 256 columns: the last workgroup of the 256 x 256
    kernel has only half a block of real weight tiles."""
    _gemm_case(K, 512, 1152, 1024, 0, 1)
    _gemm_case(K, 300, 384, 512, 0, 1, out_f32=True)


# ------------------------------------------------------------------------------------------
def _attn_case(K, Hq, Hkv, D, ctx_lens, q_lens, splits, seed=0, use_rope=True, qkv_partials=0, kv8=False):
    torch.manual_seed(seed)
    B = len(ctx_lens)
    T = sum(q_lens)
    qkv_w = (Hq + 2 * Hkv) * D
    max_len = max(c + q for c, q in zip(ctx_lens, q_lens))
    max_blocks = (max_len + 31) // 32 + 1
    num_blocks = B * max_blocks + 3
    perm = torch.randperm(num_blocks)[:B * max_blocks].reshape(B, max_blocks)
    block_table = perm.to(device=DEV, dtype=torch.int32)
    kv = torch.zeros(num_blocks, Hkv, 2, 32 * D, device=DEV, dtype=torch.uint8 if kv8 else torch.bfloat16)
    kvs = torch.zeros(num_blocks, Hkv, 2, 32, device=DEV) if kv8 else None

    def kvwrite(rows, part, S, pb, tp, ts):
        if kv8:
            K.rope_kvwrite8(rows, part, S, pb, kv, kvs, cos, sin, tp, ts, block_table, Hq, Hkv, D, use_rope)
        else:
            K.rope_kvwrite(rows, part, S, pb, kv, cos, sin, tp, ts, block_table, Hq, Hkv, D, use_rope)

    def qdq(x):
        """What the e4m3 cache holds of [n, Hkv, D] values: per (token, head) quantise + dequantise."""
        q, sc = packing.quantize_fp8(x.float().reshape(-1, D))
        return packing.dequantize_fp8(q, sc).reshape(x.shape)

    P = max_len + 8
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float64) / D))
    ang = torch.outer(torch.arange(P, dtype=torch.float64), inv)
    cos, sin = ang.cos().float().to(DEV), ang.sin().float().to(DEV)

    # history (ctx) K/V per sequence, written through the same kernel in one "prefill" call
    hist_k = [bf(torch.randn(c, Hkv, D, device=DEV)) for c in ctx_lens]
    hist_v = [bf(torch.randn(c, Hkv, D, device=DEV)) for c in ctx_lens]
    if sum(ctx_lens):
        rows, tpos, tseq = [], [], []
        for b, c in enumerate(ctx_lens):
            if c == 0:
                continue
            r = torch.zeros(c, qkv_w, device=DEV, dtype=torch.bfloat16)
            r[:, Hq * D:(Hq + Hkv) * D] = hist_k[b].reshape(c, -1)
            r[:, (Hq + Hkv) * D:] = hist_v[b].reshape(c, -1)
            rows.append(r); tpos += list(range(c)); tseq += [b] * c
        hq = torch.cat(rows)
        kvwrite(hq, None, 1, None, torch.tensor(tpos, device=DEV, dtype=torch.int32),
                torch.tensor(tseq, device=DEV, dtype=torch.int32))
    # the new chunk
    qkv = bf(torch.randn(T, qkv_w, device=DEV))
    qkv0 = qkv.clone()
    tpos, tseq, q_start = [], [], [0]
    for b, (c, q) in enumerate(zip(ctx_lens, q_lens)):
        tpos += list(range(c, c + q)); tseq += [b] * q; q_start.append(q_start[-1] + q)
    tpos_t = torch.tensor(tpos, device=DEV, dtype=torch.int32)
    if qkv_partials:
        # the same values delivered as fp32 split-K partials + bias, the bf16 buffer holding garbage
        S = qkv_partials
        pb = bf(torch.randn(qkv_w, device=DEV))
        parts = torch.randn(S, T, qkv_w, device=DEV)
        parts[0] = qkv0.float() - pb.float() - parts[1:].sum(0)
        qkv.fill_(float("nan"))
        kvwrite(qkv, parts.contiguous(), S, pb, tpos_t, torch.tensor(tseq, device=DEV, dtype=torch.int32))
    else:
        kvwrite(qkv, None, 1, None, tpos_t, torch.tensor(tseq, device=DEV, dtype=torch.int32))
    out = torch.full((T, Hq * D), float("nan"), device=DEV, dtype=torch.bfloat16)
    q_start_t = torch.tensor(q_start, device=DEV, dtype=torch.int32)
    ctx_t = torch.tensor(ctx_lens, device=DEV, dtype=torch.int32)
    max_q = max(q_lens)
    partial = None
    if splits > 1:
        n = K.attention_ws_floats(B, max_q, Hq, Hkv, D, splits)
        partial = torch.full((n,), float("nan"), device=DEV, dtype=torch.float32)
    scale = 1.0 / math.sqrt(D)
    if kv8:
        K.paged_attention8(out, qkv, kv, kvs, q_start_t, ctx_t, block_table, None, partial, max_q, Hq, Hkv, D, scale, splits)
    else:
        K.paged_attention(out, qkv, kv, q_start_t, ctx_t, block_table, None, partial, max_q, Hq, Hkv, D, scale, splits)
    torch.cuda.synchronize()

    # reference
    for b, (c, q) in enumerate(zip(ctx_lens, q_lens)):
        sl = slice(q_start[b], q_start[b + 1])
        raw = qkv0[sl]
        qn = raw[:, :Hq * D].reshape(q, Hq, D)
        kn = raw[:, Hq * D:(Hq + Hkv) * D].reshape(q, Hkv, D)
        vn = raw[:, (Hq + Hkv) * D:].reshape(q, Hkv, D)
        k_all = torch.cat([hist_k[b], kn]); v_all = torch.cat([hist_v[b], vn])
        pos_all = torch.arange(c + q, device=DEV)
        if use_rope:
            qr = bf(ref.rope(qn, cos, sin, pos_all[c:]))
            kr = bf(ref.rope(k_all, cos, sin, pos_all))
        else:
            qr, kr = qn, k_all
        # rotated q must have been written back in place
        assert_close(qkv[sl, :Hq * D].reshape(q, Hq, D), qr, 2e-2, 2e-2, "rope(q) in place")
        if kv8:
            # the cache quantises the fp32 rotation result (not its bf16 rounding) and V as given
            k_c = qdq(ref.rope(k_all, cos, sin, pos_all) if use_rope else k_all)
            v_c = qdq(v_all)
            want = ref.attention(qr, k_c, v_c, c, scale).reshape(q, Hq * D)
            assert_close(out[sl], want, 3e-2, 3e-2, f"attention over the e4m3 cache b={b} ctx={c} q={q} splits={splits}")
            nb = (c + q + 31) // 32
            idx = block_table[b, :nb].long()
            kd, vd = packing.kv8_block_to_dense(kv[idx][:, :, 0], kv[idx][:, :, 1], kvs[idx][:, :, 0], kvs[idx][:, :, 1], D)
            kd = kd.permute(0, 2, 1, 3).reshape(nb * 32, Hkv, D)[:c + q]
            vd = vd.permute(0, 2, 1, 3).reshape(nb * 32, Hkv, D)[:c + q]
            # a value on an e4m3 rounding boundary may land one step away (the two sides round the scale /
            # the rotation differently in the last bit): all within one step, nearly all identical
            for name, got, ref_c in (("K", kd, k_c), ("V", vd, v_c)):
                err = (got - ref_c).abs()
                step = ref_c.abs() * 0.126 + ref_c.abs().amax(dim=-1, keepdim=True) * 2.3e-3
                assert bool((err <= step).all()), f"{name} cache: max err {err.max().item()}"
                assert (err > 1e-6 + 1e-3 * ref_c.abs()).float().mean().item() < 0.01, f"{name} cache: too many off-by-one-step values"
            continue
        want = ref.attention(qr, kr, v_all, c, scale).reshape(q, Hq * D)
        assert_close(out[sl], want, 2e-2, 2e-2, f"attention b={b} ctx={c} q={q} Hq={Hq} Hkv={Hkv} D={D} splits={splits}")
        # cache content round-trips through the fragment layout
        nb = (c + q + 31) // 32
        slabs = kv[block_table[b, :nb].long()]                    # [nb, Hkv, 2, 32*D]
        kd, vd = packing.kv_block_to_dense(slabs[:, :, 0], slabs[:, :, 1], D)   # [nb, Hkv, 32, D]
        kd = kd.permute(0, 2, 1, 3).reshape(nb * 32, Hkv, D)[:c + q]
        vd = vd.permute(0, 2, 1, 3).reshape(nb * 32, Hkv, D)[:c + q]
        assert_close(kd, kr, 2e-2, 2e-2, "K cache")
        if qkv_partials:
            assert_close(vd, v_all, 2e-2, 2e-2, "V cache")
        else:
            assert torch.equal(vd, v_all), "V cache"


@pytest.mark.parametrize("geom", [(32, 8, 128), (32, 4, 64), (12, 12, 64), (64, 8, 128), (8, 4, 64)])
def test_attention_decode(K, geom):
    Hq, Hkv, D = geom
    _attn_case(K, Hq, Hkv, D, ctx_lens=[0, 1, 31, 32, 100, 257], q_lens=[1] * 6, splits=1)


@pytest.mark.parametrize("kv8", [False, True])
@pytest.mark.parametrize("geom", [(32, 8, 128), (32, 4, 64), (12, 12, 64), (64, 8, 128)])
def test_attention_decode_one_wave_per_item(K, geom, kv8):
    """From 1024 (sequence, kv head) items on, a decode step runs one wave per item (no cross-wave
    merge): contexts of 0 tokens, one block exactly, one token into the next block, many blocks."""
    Hq, Hkv, D = geom
    B = 1024 // Hkv + 8
    g = torch.Generator().manual_seed(Hq + D)
    ctx = torch.randint(0, 200, (B,), generator=g).tolist()
    ctx[:6] = [0, 1, 31, 32, 33, 64]
    ctx[-1] = 700
    _attn_case(K, Hq, Hkv, D, ctx_lens=ctx, q_lens=[1] * B, splits=1, kv8=kv8)


@pytest.mark.parametrize("workgroups", [64, 256])
@pytest.mark.parametrize("geom", [(32, 8, 128), (32, 4, 64)])
def test_attention_decode_persistent_workgroups(K, geom, workgroups, monkeypatch):
    """The experiment switch DLIA_ATTN_PERSIST: the one-wave-per-item body in a fixed number of 4-wave
    workgroups that walk the item list with a stride (item counts that are not multiples of the stride,
    workgroups whose last turn has fewer than four items)."""
    Hq, Hkv, D = geom
    monkeypatch.setenv("DLIA_ATTN_PERSIST", str(workgroups))
    B = 1024 // Hkv + 7
    g = torch.Generator().manual_seed(Hq + D + workgroups)
    ctx = torch.randint(0, 200, (B,), generator=g).tolist()
    ctx[:6] = [0, 1, 31, 32, 33, 64]
    ctx[-1] = 700
    _attn_case(K, Hq, Hkv, D, ctx_lens=ctx, q_lens=[1] * B, splits=1)


def test_attention_decode_one_wave_per_item_with_split_kv(K):
    ctx = [5, 700, 64, 333] * 10 + [1, 31]
    _attn_case(K, 32, 8, 128, ctx_lens=ctx, q_lens=[1] * len(ctx), splits=4)          # 42 x 8 x 4 = 1344 items
    _attn_case(K, 32, 8, 128, ctx_lens=ctx, q_lens=[1] * len(ctx), splits=4, kv8=True)


@pytest.mark.parametrize("splits", [2, 4, 8])
def test_attention_decode_splitkv(K, splits):
    _attn_case(K, 32, 8, 128, ctx_lens=[5, 700, 64, 333], q_lens=[1] * 4, splits=splits)
    _attn_case(K, 32, 4, 64, ctx_lens=[1000, 3], q_lens=[1, 1], splits=splits)


@pytest.mark.parametrize("geom", [(32, 8, 128), (32, 4, 64), (12, 12, 64)])
def test_attention_prefill_ragged(K, geom):
    Hq, Hkv, D = geom
    _attn_case(K, Hq, Hkv, D, ctx_lens=[0, 0, 40, 96], q_lens=[77, 5, 33, 64], splits=1)


@pytest.mark.parametrize("kv8", [False, True])
@pytest.mark.parametrize("geom", [(32, 8, 128), (32, 4, 64), (12, 12, 64), (64, 8, 128), (8, 4, 64)])
def test_attention_prefill_long_chunks(K, geom, kv8):
    """Chunks of more than 128 query columns per sequence run on the prefill-shaped kernel (waves split the
    query columns, KV blocks shared through LDS): several query tiles per sequence, a cached context of many
    blocks (most of them on the unmasked path), a sequence shorter than one tile, MHA and GQA groups of 2-8."""
    Hq, Hkv, D = geom
    _attn_case(K, Hq, Hkv, D, ctx_lens=[0, 300, 37], q_lens=[300, 150, 3], splits=1, kv8=kv8)
    _attn_case(K, Hq, Hkv, D, ctx_lens=[1000], q_lens=[129], splits=1, seed=3, kv8=kv8)
    # two sequences: 8 / 16 / 24 (sequence, kv head) groups for every geometry, i.e. the workgroup order that puts
    # group i on XCD i % 8 (the cases above have 12, 4 or 36 groups for the 64-wide heads: 8 / groups XCDs per
    # group, or the plain order); `out` starts as NaN, so a tile the mapping missed would show
    _attn_case(K, Hq, Hkv, D, ctx_lens=[10, 200], q_lens=[140, 260], splits=1, seed=4, kv8=kv8)
    # 600 tokens in one call: from 512 tokens up the KV write runs one workgroup per token for every geometry
    # (below that a token's heads are spread over two); cache contents are compared exactly
    _attn_case(K, Hq, Hkv, D, ctx_lens=[0], q_lens=[600], splits=1, seed=6, kv8=kv8)


@pytest.mark.parametrize("kv8", [False, True])
@pytest.mark.parametrize("geom", [(24, 8, 128), (28, 4, 128), (12, 2, 128), (40, 8, 128), (14, 2, 64)])
def test_attention_gqa_groups_that_are_not_powers_of_two(K, geom, kv8):
    """Groups of 3 (Llama-3.2-3B), 7 (Qwen2.5-7B / 0.5B), 6 and 5 query heads per kv head: a position takes the next
    power of two of query columns and the surplus columns are padding. Every kernel shape: decode with four waves per
    item, with one wave per item (>= 1024 items), split KV with the combine kernel, ragged chunks (two column tiles per
    workgroup), chunks on the prefill-shaped kernel. `out` starts as NaN: a head nobody wrote would show."""
    Hq, Hkv, D = geom
    _attn_case(K, Hq, Hkv, D, ctx_lens=[0, 1, 31, 32, 100, 257], q_lens=[1] * 6, splits=1, kv8=kv8)
    B = 1024 // Hkv + 3
    ctx = torch.randint(0, 150, (B,), generator=torch.Generator().manual_seed(Hq)).tolist()
    ctx[:4] = [0, 31, 32, 33]
    _attn_case(K, Hq, Hkv, D, ctx_lens=ctx, q_lens=[1] * B, splits=1, seed=1, kv8=kv8)
    _attn_case(K, Hq, Hkv, D, ctx_lens=[5, 700, 64, 333], q_lens=[1] * 4, splits=4, seed=2, kv8=kv8)
    _attn_case(K, Hq, Hkv, D, ctx_lens=[0, 0, 40, 96], q_lens=[77, 5, 33, 64], splits=1, seed=3, kv8=kv8)
    _attn_case(K, Hq, Hkv, D, ctx_lens=[130, 0], q_lens=[20, 50], splits=2, seed=4, kv8=kv8)
    _attn_case(K, Hq, Hkv, D, ctx_lens=[0, 300, 37], q_lens=[300, 150, 3], splits=1, seed=5, kv8=kv8)


def test_attention_prefill_chunk_splitkv_norope(K):
    _attn_case(K, 12, 12, 64, ctx_lens=[130, 0], q_lens=[20, 50], splits=2, use_rope=False)


@pytest.mark.parametrize("geom", [(32, 8, 128), (32, 4, 64), (12, 12, 64), (64, 8, 128)])
def test_attention_fp8_kv_cache_decode(K, geom):
    """e4m3 KV cache with per-(token, head) scales: write kernel + attention against the fp32 reference
    over the dequantised cache contents."""
    Hq, Hkv, D = geom
    _attn_case(K, Hq, Hkv, D, ctx_lens=[0, 1, 31, 32, 100, 257], q_lens=[1] * 6, splits=1, kv8=True)


def test_attention_fp8_kv_cache_prefill_splitkv_and_partials(K):
    _attn_case(K, 32, 8, 128, ctx_lens=[0, 0, 40, 96], q_lens=[77, 5, 33, 64], splits=1, kv8=True)
    _attn_case(K, 32, 8, 128, ctx_lens=[5, 700, 64, 333], q_lens=[1] * 4, splits=4, kv8=True)
    _attn_case(K, 12, 12, 64, ctx_lens=[130, 0], q_lens=[20, 50], splits=2, use_rope=False, kv8=True)
    _attn_case(K, 32, 8, 128, ctx_lens=[3, 64], q_lens=[1, 1], splits=1, qkv_partials=4, kv8=True)


@pytest.mark.parametrize("use_rope", [True, False])
def test_rope_kvwrite_from_split_k_partials(K, use_rope):
    _attn_case(K, 8, 4, 64, ctx_lens=[3, 40], q_lens=[1, 1], splits=1, qkv_partials=3, use_rope=use_rope)
    _attn_case(K, 32, 8, 128, ctx_lens=[0, 17], q_lens=[9, 30], splits=1, qkv_partials=2, use_rope=use_rope)


def test_attention_softmax_rescale_spike(K):
    """Force the online-softmax rescale: late keys with much larger scores than early ones."""
    torch.manual_seed(5)
    _attn_case(K, 8, 4, 64, ctx_lens=[200], q_lens=[1], splits=1, seed=11)


# ------------------------------------------------------------------------------------------
def _sample(K, logits, temp, top_k, top_p, seeds, steps, vocab, cap=128):
    R = logits.shape[0]
    out = torch.full((R,), -1, device=DEV, dtype=torch.int32)
    probs = torch.zeros(R, cap, device=DEV, dtype=torch.float32)
    ids = torch.zeros(R, cap, device=DEV, dtype=torch.int32)
    K.sample(out, logits, torch.tensor(temp, device=DEV, dtype=torch.float32),
             torch.tensor(top_k, device=DEV, dtype=torch.int32), torch.tensor(top_p, device=DEV, dtype=torch.float32),
             torch.tensor(seeds, device=DEV, dtype=torch.int64), torch.tensor(steps, device=DEV, dtype=torch.int32),
             vocab, probs, ids)
    torch.cuda.synchronize()
    return out.cpu(), probs.cpu(), ids.cpu()


@pytest.mark.parametrize("shape", [(300, 1000, 256), (520, 50257, 128), (1024, 128256, 256), (700, 32000, 384),
                                   (1, 128256, 256), (7, 1000, 256), (40, 50257, 128), (128, 32000, 384),
                                   (129, 1000, 256), (200, 50257, 128), (256, 128256, 256)])
def test_gemm_head_writes_logits_and_chunk_maxima(K, shape):
    """The lm_head above 256 rows (tile kernel) and up to 256 rows (weight-streaming kernels, no split): fp32 logits
    equal to the plain GEMM's (and to an fp32 torch matmul), plus, per
    row, the maximum of every 32-column chunk over the real vocabulary — padded columns (zero weights,
    logit 0) excluded, also when the vocabulary ends inside a chunk or inside a 256-column block. With a
    split-K workspace too (thin shapes)."""
    M, V, Kd = shape
    torch.manual_seed(2)
    x = bf(torch.randn(M, Kd, device=DEV))
    w = bf(torch.randn(V, Kd, device=DEV) / math.sqrt(Kd)) - 0.05        # mostly negative logits: a padded 0 would win a chunk
    w = bf(w)
    wp = packing.pack_weight(w)
    n_pad = packing.padded_n(V)
    want = x.float() @ w.float().t()
    for ws in (None, torch.zeros(int(K.gemm_ws_floats(M, n_pad, Kd)) + 4096, device=DEV)):
        logits = torch.full((M, n_pad), 7.0, device=DEV)
        cm = torch.full((M, (n_pad + 255) // 256 * 8), 9.0, device=DEV)
        K.gemm_head(logits, cm, x, wp, ws, n_pad, V)
        plain = torch.zeros(M, n_pad, device=DEV)
        K.gemm(plain, x, wp, None, None, None, n_pad, 0, 1 if M <= 256 else 0, 2 if M <= 128 else 0)
        torch.cuda.synchronize()
        assert torch.equal(logits[:, :V], plain[:, :V])
        assert_close(logits[:, :V], want, 2e-2, 2e-2, "logits")
        nch = (V + 31) // 32
        padded = torch.full((M, nch * 32), float("-inf"), device=DEV)
        padded[:, :V] = logits[:, :V]
        assert torch.equal(cm[:, :nch], padded.view(M, nch, 32).amax(2)), "chunk maxima"


def _chunk_max(logits, V):
    R = logits.shape[0]
    nch = (V + 31) // 32
    padded = torch.full((R, nch * 32), float("-inf"), device=logits.device)
    padded[:, :V] = logits[:, :V]
    return padded.view(R, nch, 32).amax(2).contiguous()


@pytest.mark.parametrize("V", [32000, 50257, 128256, 151936])
def test_guided_sampler_draws_the_same_tokens_as_the_whole_row_sampler(K, V):
    """With the chunk maxima the sampler reads only the chunks that can hold a candidate; the tokens must
    be those of the whole-row method for every rule: k = 1 / 5 / 50 / 1024, greedy, rows of exact ties
    (at the k-th value too), a constant row (every chunk reaches the bound: falls back), a row whose
    largest values all sit in ONE chunk, and k <= 0 (no top-k: whole-row path either way)."""
    torch.manual_seed(11)
    R = 12
    ld = (V + 127) // 128 * 128
    logits = torch.randn(R, ld, device=DEV) * 3
    logits[:, V:] = 1e9
    logits[2, :V] = (logits[2, :V] * 2).round() / 2                    # heavy ties
    logits[3, :V] = 0.25                                              # constant row
    logits[4, 64:96] += 40.0                                          # the whole top-k in one chunk
    logits[5, :V] = (logits[5, :V]).round()                            # integer logits: ties at the k-th value
    logits[6, V - 1] = 50.0                                           # the maximum in the last, partial chunk
    temps = [0.7, 1.0, 0.7, 0.9, 0.7, 0.7, 0.0, 0.0, 1.2, 0.7, 0.7, 0.5]
    ks = [50, 50, 50, 50, 20, 50, 50, 50, 1, 1024, 0, 5]
    ps = [0.9, 0.9, 1.0, 1.0, 0.9, 1.0, 0.9, 0.9, 0.9, 0.95, 0.9, 0.3]
    seeds, steps = list(range(21, 21 + R)), [3] * R
    cm = _chunk_max(logits, V)
    for trial in range(3):
        steps = [trial] * R
        want, wp_, wi_ = _sample(K, logits, temps, ks, ps, seeds, steps, V)
        out = torch.full((R,), -1, device=DEV, dtype=torch.int32)
        probs = torch.zeros(R, 128, device=DEV)
        ids = torch.zeros(R, 128, device=DEV, dtype=torch.int32)
        t = lambda v, dt: torch.tensor(v, device=DEV, dtype=dt)
        K.sample_guided(out, logits, cm, t(temps, torch.float32), t(ks, torch.int32), t(ps, torch.float32), t(seeds, torch.int64),
                        t(steps, torch.int32), V, probs, ids)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), want), (trial, out.cpu().tolist(), want.tolist())
        sorted_rows = [r for r in range(R) if ks[r] > 0 and temps[r] > 0]
        assert torch.equal(ids.cpu()[sorted_rows], wi_[sorted_rows]) and torch.equal(probs.cpu()[sorted_rows], wp_[sorted_rows])
    assert want[6] == V - 1 and want[7] == logits[7, :V].argmax().item()


@pytest.mark.parametrize("V", [320, 32000, 50257, 128256, 151936])
def test_sampler_matches_reference_filter(K, V):
    torch.manual_seed(3)
    R = 6
    ld = (V + 127) // 128 * 128
    logits = torch.randn(R, ld, device=DEV) * 3
    logits[:, V:] = 1e9          # padded columns must be ignored
    logits[2, :V] = (logits[2, :V] * 4).round() / 4       # many exact ties
    temps = [0.7, 1.0, 0.7, 1.3, 0.7, 0.2]
    ks = [50, 50, 50, 5, 1, 50]
    # row 2 (exact ties) uses top_p = 1: which of several EQUAL logits falls inside the nucleus
    # depends on the sort's tie order, which the reference leaves to torch.sort's implementation
    ps = [0.9, 0.9, 1.0, 0.5, 0.9, 1.0]
    out, probs, ids = _sample(K, logits, temps, ks, ps, [1, 2, 3, 4, 5, 6], [0] * R, V)
    for r in range(R):
        want = filter_logits(logits[r, :V].cpu(), temps[r], ks[r], ps[r])
        nz = ids[r] >= 0
        got = torch.zeros(V)
        got[ids[r][nz].long()] = probs[r][nz]
        # candidate sets may differ only where the cumulative mass sits within rounding of top_p
        assert (got - want).abs().max() < 2e-3, f"row {r}: filtered distribution differs"
        assert want[out[r]] > 0, "sampled token must be in the reference support"
        assert 0 <= out[r] < V


def test_sampler_greedy_and_determinism(K):
    torch.manual_seed(4)
    V = 32000
    logits = torch.randn(4, V, device=DEV)
    out, _, _ = _sample(K, logits, [0.0, -1.0, 0.7, 0.7], [50] * 4, [0.9] * 4, [7, 7, 7, 7], [0, 0, 3, 3], V)
    assert out[0] == logits[0].argmax().item() and out[1] == logits[1].argmax().item()
    out2, _, _ = _sample(K, logits, [0.0, -1.0, 0.7, 0.7], [50] * 4, [0.9] * 4, [7, 7, 7, 7], [0, 0, 3, 3], V)
    assert torch.equal(out, out2), "same (seed, step) must give the same draw"
    alone, _, _ = _sample(K, logits[2:3].contiguous(), [0.7], [50], [0.9], [7], [3], V)
    assert alone[0] == out[2], "a draw must not depend on the row the request occupies"


@pytest.mark.parametrize("guided", [False, True])
def test_sample_step_draws_like_sample_and_does_the_steps_bookkeeping(K, guided):
    """`sample_step`: same tokens as `sample` / `sample_guided`, and the thread that writes a row's token
    also copies it into the next step's ids and advances the row's draw counter and context length —
    over the greedy, top-k and full-row paths."""
    torch.manual_seed(12)
    V, R = 32000, 6
    logits = torch.randn(R, V + 64, device=DEV)
    t = lambda v, dt: torch.tensor(v, device=DEV, dtype=dt)
    temps, ks, ps = t([0.0, 0.7, 0.7, 1.2, 0.9, 0.0], torch.float32), t([50, 50, 0, 1000, 5, 1], torch.int32), t([0.9, 0.9, 0.8, 1.0, 0.5, 1.0], torch.float32)
    seeds, steps0, ctx0 = t([3, 4, 5, 6, 7, 8], torch.int64), t([0, 9, 2, 5, 1, 0], torch.int32), t([10, 0, 31, 32, 500, 7], torch.int32)
    cm = _chunk_max(logits, V) if guided else None
    want = torch.full((R,), -1, device=DEV, dtype=torch.int32)
    if guided:
        K.sample_guided(want, logits, cm, temps, ks, ps, seeds, steps0, V, None, None)
    else:
        K.sample(want, logits, temps, ks, ps, seeds, steps0, V, None, None)
    out, ids = torch.full((R,), -1, device=DEV, dtype=torch.int32), torch.full((R,), -1, device=DEV, dtype=torch.int32)
    steps, ctx = steps0.clone(), ctx0.clone()
    K.sample_step(out, logits, cm, temps, ks, ps, seeds, steps, V, ids, ctx)
    assert torch.equal(out, want) and torch.equal(ids, want)
    assert torch.equal(steps, steps0 + 1) and torch.equal(ctx, ctx0 + 1)
    K.sample_step(out, logits, cm, temps, ks, ps, seeds, steps, V, None, None)       # no ids, no ctx: only the counter moves
    assert torch.equal(steps, steps0 + 2) and torch.equal(ctx, ctx0 + 1)


def test_sampler_distribution(K):
    """Empirical frequencies of many independent draws follow the filtered distribution."""
    torch.manual_seed(6)
    V, R = 1000, 4096
    base = torch.randn(V, device=DEV) * 2
    logits = base.unsqueeze(0).repeat(R, 1).contiguous()
    out, _, _ = _sample(K, logits, [0.8] * R, [20] * R, [0.95] * R, list(range(R)), [0] * R, V)
    want = filter_logits(base.cpu(), 0.8, 20, 0.95)
    freq = torch.bincount(out.long(), minlength=V).float() / R
    assert (freq[want == 0] == 0).all()
    assert (freq - want).abs().max() < 0.04


@pytest.mark.parametrize("mode", ["rms", "layer"])
def test_norm_with_quantised_output(K, mode):
    """norm_q = norm followed by the per-row e4m3 quantiser, in one launch (incl. the partial fold)."""
    torch.manual_seed(3)
    M, H, S = 24, 1024, 4
    x0 = bf(torch.randn(M, H, device=DEV) * 3)
    g = bf(torch.rand(H, device=DEV) + 0.5)
    b = bf(torch.randn(H, device=DEV) * 0.1) if mode == "layer" else None
    part = torch.randn(S, M, H, device=DEV) * 0.3
    pb = bf(torch.randn(H, device=DEV) * 0.1)
    m = 0 if mode == "rms" else 1
    xa, xb = x0.clone(), x0.clone()
    want = torch.empty_like(x0)
    K.norm(want, xa, None, part.reshape(-1), S, pb, g, b, None, 1e-5, m)
    q = torch.zeros(M, 2048, device=DEV, dtype=torch.uint8)[:, :H]
    sc = torch.zeros(M, device=DEV)
    K.norm_q(q, sc, xb, None, part.reshape(-1), S, pb, g, b, 1e-5, m)
    assert torch.equal(xa, xb), "residual stream must be updated identically"
    deq = packing.dequantize_fp8(q.contiguous(), sc)
    w = want.float()
    assert_close(sc, w.abs().amax(1) / 448.0, 2e-2, 1e-6, "row scale")       # bf16 rounding of `want` only
    assert ((deq - w).abs() <= 0.07 * w.abs() + 0.004 * w.abs().amax(1, keepdim=True)).all()
