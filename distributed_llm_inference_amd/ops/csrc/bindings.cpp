// torch custom-op bindings for the HIP kernel library (namespace `dlia`).
// Every op validates shapes / dtypes / devices on the host before launching: a kernel that
// indexes out of bounds can take the whole GPU down, so nothing reaches a launch unchecked.
// All launches go to PyTorch's current HIP stream, which makes them capturable in a hipGraph
// through torch.cuda.graph().
#include <torch/library.h>
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/util/Optional.h>

#include <atomic>
#include <cstdlib>
#include "launch.h"

extern "C" long dlia_paged_attention_ws_floats(int B, int max_q_len, int Hq, int Hkv, int D, int splits);

namespace {

using at::Tensor;
using OptTensor = const std::optional<Tensor>&;

inline hipStream_t cur_stream() { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

#define DLIA_CHECK(cond, ...) TORCH_CHECK(cond, "dlia: ", __VA_ARGS__)
// HIP's current device (and with it PyTorch's current stream) is per THREAD: a helper thread of a rank that serves
// cuda:3 starts on device 0. Every op therefore runs on the device of its output tensor, whatever the calling thread's is.
#define DLIA_ON_DEVICE_OF(t) \
  DLIA_CHECK((t).is_cuda(), #t " must be a GPU tensor"); \
  const c10::hip::HIPGuardMasqueradingAsCUDA dlia_device_guard_((t).device())

inline void check_dev(const Tensor& t, const char* name) {
  DLIA_CHECK(t.is_cuda(), name, " must be a GPU tensor");
}
inline void check_bf16(const Tensor& t, const char* name) {
  check_dev(t, name);
  DLIA_CHECK(t.scalar_type() == at::kBFloat16, name, " must be bfloat16");
}
inline void check_i32(const Tensor& t, const char* name) {
  check_dev(t, name);
  DLIA_CHECK(t.scalar_type() == at::kInt && t.is_contiguous(), name, " must be contiguous int32");
}
inline void check_f32(const Tensor& t, const char* name) {
  check_dev(t, name);
  DLIA_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous(), name, " must be contiguous float32");
}
inline void check_rows(const Tensor& t, const char* name) {
  DLIA_CHECK(t.dim() == 2 && t.stride(1) == 1, name, " must be 2-D with unit inner stride");
  DLIA_CHECK(t.stride(0) % 8 == 0, name, " row stride must be a multiple of 8 elements (16-byte rows)");
  DLIA_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name, " must be 16-byte aligned");
}

// out[r,:] = norm(x[row_idx[r] or r,:] (+addend -> x)) * w (+ b); mode 0 rms, 1 layernorm, 2 fold only.
static void norm_impl(Tensor out, Tensor x, OptTensor delta, OptTensor partial, int64_t splits, OptTensor pbias,
                      const Tensor& w, OptTensor b, OptTensor row_idx, double eps, int64_t mode, OptTensor qscale) {
  DLIA_ON_DEVICE_OF(out);
  check_bf16(x, "x"); check_bf16(w, "w");
  check_rows(out, "out"); check_rows(x, "x");
  float* qs = nullptr;
  if (qscale.has_value()) {
    check_f32(*qscale, "qscale");
    DLIA_CHECK(out.scalar_type() == at::kByte && out.stride(0) % 16 == 0, "quantised norm output must be uint8 rows");
    DLIA_CHECK(mode != 2, "fold-only norm has no output to quantise");
    qs = qscale->data_ptr<float>();
  } else {
    check_bf16(out, "out");
  }
  const int64_t H = x.size(1);
  DLIA_CHECK(H % 8 == 0, "hidden size must be a multiple of 8");
  DLIA_CHECK(mode >= 0 && mode <= 2, "norm mode");
  DLIA_CHECK(out.size(1) == H && w.numel() == H && w.is_contiguous(), "norm: shape mismatch");
  const int* ridx = nullptr;
  int64_t rows = x.size(0);
  if (row_idx.has_value()) {
    check_i32(*row_idx, "row_idx");
    rows = row_idx->numel();
    ridx = row_idx->data_ptr<int>();
  }
  DLIA_CHECK(out.size(0) >= rows, "norm: out has too few rows");
  if (qs) DLIA_CHECK(qscale->is_contiguous() && qscale->numel() >= rows, "qscale shape");
  const void* dp = nullptr;
  if (delta.has_value()) {
    check_bf16(*delta, "delta"); check_rows(*delta, "delta");
    DLIA_CHECK(delta->sizes() == x.sizes() && delta->stride(0) == x.stride(0), "delta must match x");
    dp = delta->data_ptr();
  }
  const float* pp = nullptr; const void* pb = nullptr; int64_t pM = x.size(0);
  if (partial.has_value()) {
    DLIA_CHECK(!delta.has_value(), "norm: delta and partial are exclusive");
    check_f32(*partial, "partial");
    DLIA_CHECK(splits >= 1 && partial->numel() >= splits * pM * H, "norm: partial workspace smaller than [splits, rows, H]");
    pp = partial->data_ptr<float>();
    if (pbias.has_value()) {
      check_bf16(*pbias, "pbias");
      DLIA_CHECK(pbias->numel() == H && pbias->is_contiguous(), "pbias shape");
      pb = pbias->data_ptr();
    }
  }
  DLIA_CHECK(mode != 2 || pp || dp, "norm mode 2 needs an addend");
  const void* bp = nullptr;
  if (mode == 1) {
    DLIA_CHECK(b.has_value(), "layernorm needs a bias");
    check_bf16(*b, "b");
    DLIA_CHECK(b->numel() == H && b->is_contiguous(), "bias shape");
    bp = b->data_ptr();
  }
  dlia_norm(out.data_ptr(), x.data_ptr(), dp, pp, (int)splits, pM, pb, w.data_ptr(), bp, ridx, (int)rows, (int)H,
            x.stride(0), out.stride(0), (float)eps, (int)mode, qs, cur_stream());
}

void norm(Tensor out, Tensor x, OptTensor delta, OptTensor partial, int64_t splits, OptTensor pbias,
          const Tensor& w, OptTensor b, OptTensor row_idx, double eps, int64_t mode) {
  norm_impl(out, x, delta, partial, splits, pbias, w, b, row_idx, eps, mode, c10::nullopt);
}

// norm whose output is e4m3 bytes + per-row scale (input of gemm_a8w8)
void norm_q(Tensor out_q, Tensor qscale, Tensor x, OptTensor delta, OptTensor partial, int64_t splits, OptTensor pbias,
            const Tensor& w, OptTensor b, double eps, int64_t mode) {
  norm_impl(out_q, x, delta, partial, splits, pbias, w, b, c10::nullopt, eps, mode, qscale);
}

void embed(Tensor out, const Tensor& ids, OptTensor pos, const Tensor& wte, OptTensor wpe) {
  DLIA_ON_DEVICE_OF(out);
  check_bf16(out, "out"); check_bf16(wte, "wte"); check_i32(ids, "ids");
  DLIA_CHECK(out.is_contiguous() && wte.is_contiguous() && wte.dim() == 2, "embed: contiguous 2-D tables");
  const int64_t T = ids.numel(), H = wte.size(1);
  DLIA_CHECK(out.dim() == 2 && out.size(0) >= T && out.size(1) == H && H % 8 == 0, "embed: out shape");
  const void* wp = nullptr; const int* pp = nullptr; int maxp = 0;
  if (wpe.has_value()) {
    check_bf16(*wpe, "wpe");
    DLIA_CHECK(pos.has_value(), "wpe needs positions");
    check_i32(*pos, "pos");
    DLIA_CHECK(pos->numel() >= T && wpe->is_contiguous() && wpe->size(1) == H, "embed: wpe/pos shape");
    wp = wpe->data_ptr(); pp = pos->data_ptr<int>(); maxp = (int)wpe->size(0);
  }
  dlia_embed(out.data_ptr(), ids.data_ptr<int>(), pp, wte.data_ptr(), wp, (int)T, (int)H, (int)wte.size(0),
             maxp, cur_stream());
}

struct KvGeom { int64_t num_blocks; };
inline KvGeom check_kv(const Tensor& kv, int64_t Hkv, int64_t D) {
  check_bf16(kv, "kv_cache");
  DLIA_CHECK(kv.is_contiguous() && kv.dim() == 4 && kv.size(1) == Hkv && kv.size(2) == 2 && kv.size(3) == 32 * D,
             "kv_cache must be [num_blocks, Hkv, 2, 32*D]");
  return {kv.size(0)};
}

// e4m3 cache: bytes [num_blocks, Hkv, 2, 32*D] + fp32 scales [num_blocks, Hkv, 2, 32]
inline KvGeom check_kv8(const Tensor& kv, const Tensor& kv_scale, int64_t Hkv, int64_t D) {
  check_dev(kv, "kv_cache"); check_f32(kv_scale, "kv_scale");
  DLIA_CHECK(kv.scalar_type() == at::kByte, "fp8 kv_cache must be uint8");
  DLIA_CHECK(kv.is_contiguous() && kv.dim() == 4 && kv.size(1) == Hkv && kv.size(2) == 2 && kv.size(3) == 32 * D,
             "fp8 kv_cache must be [num_blocks, Hkv, 2, 32*D] bytes");
  DLIA_CHECK(kv_scale.is_contiguous() && kv_scale.dim() == 4 && kv_scale.size(0) == kv.size(0) && kv_scale.size(1) == Hkv &&
             kv_scale.size(2) == 2 && kv_scale.size(3) == 32, "kv_scale must be [num_blocks, Hkv, 2, 32]");
  DLIA_CHECK(reinterpret_cast<uintptr_t>(kv.data_ptr()) % 16 == 0 && reinterpret_cast<uintptr_t>(kv_scale.data_ptr()) % 16 == 0,
             "kv cache alignment");
  return {kv.size(0)};
}

static void rope_kvwrite_impl(Tensor qkv, OptTensor partial, int64_t splits, OptTensor pbias, Tensor kv_cache,
                              OptTensor kv_scale, const Tensor& cos_t, const Tensor& sin_t,
                              const Tensor& tok_pos, const Tensor& tok_seq, const Tensor& block_table,
                              int64_t Hq, int64_t Hkv, int64_t D, bool use_rope) {
  DLIA_ON_DEVICE_OF(qkv);
  check_bf16(qkv, "qkv");
  DLIA_CHECK(qkv.dim() == 2 && qkv.is_contiguous() && qkv.size(1) == (Hq + 2 * Hkv) * D, "qkv must be [T,(Hq+2Hkv)*D]");
  DLIA_CHECK(D == 64 || D == 128, "head_dim must be 64 or 128");
  const int64_t T = qkv.size(0);
  auto g = kv_scale.has_value() ? check_kv8(kv_cache, *kv_scale, Hkv, D) : check_kv(kv_cache, Hkv, D);
  check_i32(tok_pos, "tok_pos"); check_i32(tok_seq, "tok_seq"); check_i32(block_table, "block_table");
  DLIA_CHECK(tok_pos.numel() >= T && tok_seq.numel() >= T && block_table.dim() == 2, "rope_kvwrite: metadata shapes");
  int rope_len = 1;
  const float* cp = nullptr; const float* sp = nullptr;
  if (use_rope) {
    check_f32(cos_t, "cos"); check_f32(sin_t, "sin");
    DLIA_CHECK(cos_t.dim() == 2 && cos_t.size(1) == D / 2 && sin_t.sizes() == cos_t.sizes(), "rope tables [P, D/2]");
    rope_len = (int)cos_t.size(0); cp = cos_t.data_ptr<float>(); sp = sin_t.data_ptr<float>();
  }
  const float* pp = nullptr; const void* pb = nullptr;
  if (partial.has_value()) {
    check_f32(*partial, "partial");
    DLIA_CHECK(splits >= 1 && partial->numel() >= splits * T * qkv.size(1), "rope: partial workspace smaller than [splits, T, W]");
    pp = partial->data_ptr<float>();
    if (pbias.has_value()) {
      check_bf16(*pbias, "pbias");
      DLIA_CHECK(pbias->numel() == qkv.size(1) && pbias->is_contiguous(), "pbias shape");
      pb = pbias->data_ptr();
    }
  }
  // tok_seq values index block_table rows: the caller guarantees tok_seq < block_table.size(0)
  dlia_rope_kvwrite(qkv.data_ptr(), pp, (int)splits, pb, kv_cache.data_ptr(),
                    kv_scale.has_value() ? kv_scale->data_ptr<float>() : nullptr, cp, sp, tok_pos.data_ptr<int>(),
                    tok_seq.data_ptr<int>(), block_table.data_ptr<int>(), (int)T, (int)Hq, (int)Hkv, (int)D,
                    (int)block_table.size(1), (int)g.num_blocks, rope_len, use_rope ? 1 : 0, cur_stream());
}

void rope_kvwrite(Tensor qkv, OptTensor partial, int64_t splits, OptTensor pbias, Tensor kv_cache,
                  const Tensor& cos_t, const Tensor& sin_t,
                  const Tensor& tok_pos, const Tensor& tok_seq, const Tensor& block_table,
                  int64_t Hq, int64_t Hkv, int64_t D, bool use_rope) {
  rope_kvwrite_impl(qkv, partial, splits, pbias, kv_cache, c10::nullopt, cos_t, sin_t, tok_pos, tok_seq, block_table, Hq, Hkv, D, use_rope);
}

// The same into the e4m3 cache: quantises K (after RoPE) and V per (token, kv head).
void rope_kvwrite8(Tensor qkv, OptTensor partial, int64_t splits, OptTensor pbias, Tensor kv_cache, Tensor kv_scale,
                   const Tensor& cos_t, const Tensor& sin_t,
                   const Tensor& tok_pos, const Tensor& tok_seq, const Tensor& block_table,
                   int64_t Hq, int64_t Hkv, int64_t D, bool use_rope) {
  rope_kvwrite_impl(qkv, partial, splits, pbias, kv_cache, kv_scale, cos_t, sin_t, tok_pos, tok_seq, block_table, Hq, Hkv, D, use_rope);
}

// Kernel for bf16 GEMMs of more than 256 rows: 0 = default (256 x 256 ping-pong tiles, gemm_pingpong.hip),
// 2 = the 256 x 128 loader-wave kernel walking 256-row blocks (gemm_rows.hip), 4 = the same with 128- / 64-row
// blocks on shapes with few column blocks (this one also applies to the fp8 x fp8 GEMM). Starts from
// DLIA_GEMM_LARGE=stream; the `large_gemm_kernel` op changes it for A/B runs and tests. Returns the old value.
// floats of workspace a bf16 GEMM of this shape wants (0: none): the in-launch split-K of the large-M kernel
int64_t gemm_ws_floats(int64_t M, int64_t N, int64_t K) {
  return M > 256 ? dlia_gemm_pingpong_ws_floats((int)M, (int)N, (int)K) : 0;
}

int64_t large_gemm_kernel(int64_t which) {
  DLIA_CHECK(which == 0 || which == 2 || which == 4 || which == 6,
             "large_gemm_kernel: 0 ping-pong (default), 2 stream, 4 stream with narrow row blocks, 6 four-wave 256 x 256");
  const int narrow_before = dlia_gemm_rows_set_narrow(which == 4 ? 1 : 0);
  const int before = dlia_gemm_large_choice(which == 4 ? 2 : (which == 6 ? 4 : (int)which));
  return before == 4 ? 6 : (before == 2 && narrow_before ? 4 : before);
}

// out[M, N_out] = epi(x[M,K] @ W^T); wp = packed weights (uint8 view of the packed image).
// epi: 0 none, 1 +bias, 2 +residual, 3 +bias+residual, 4 swiglu, 5 +bias -> gelu_tanh.
// wscale given: wp holds fp8 e4m3 weights (1 byte each) and wscale one fp32 scale per packed row.
static void gemm_impl(Tensor out, const Tensor& x, const Tensor& wp, OptTensor wscale, OptTensor bias,
                      OptTensor residual, OptTensor workspace, int64_t N, int64_t epi, int64_t splitk, int64_t ntw) {
  DLIA_ON_DEVICE_OF(out);
  check_bf16(x, "x"); check_rows(x, "x"); check_dev(out, "out"); check_dev(wp, "wp");
  const int64_t M = x.size(0), K = x.size(1);
  DLIA_CHECK(K % 128 == 0, "K must be a multiple of 128");
  DLIA_CHECK(N % 4 == 0, "N must be a multiple of 4");
  DLIA_CHECK(epi >= 0 && epi <= 6, "bad epilogue");
  const int64_t n_pad = (N + 127) / 128 * 128;
  const bool w8 = wscale.has_value();
  const int64_t wbytes = n_pad * K * (w8 ? 1 : 2);
  DLIA_CHECK(wp.is_contiguous() && wp.numel() * wp.element_size() == wbytes,
             "packed weight has ", wp.numel() * wp.element_size(), " bytes, expected ", wbytes);
  const float* wsc = nullptr;
  if (w8) {
    check_f32(*wscale, "wscale");
    DLIA_CHECK(wscale->is_contiguous() && wscale->numel() == n_pad, "wscale must hold one scale per packed row (", n_pad, ")");
    DLIA_CHECK(reinterpret_cast<uintptr_t>(wscale->data_ptr()) % 16 == 0, "wscale alignment");
    wsc = wscale->data_ptr<float>();
  }
  DLIA_CHECK(reinterpret_cast<uintptr_t>(wp.data_ptr()) % 16 == 0, "packed weight alignment");
  const int64_t n_out = epi == 4 ? N / 2 : N;
  if (epi == 4) DLIA_CHECK(N % 32 == 0, "swiglu needs gate/up tile pairs");
  const bool f32 = out.scalar_type() == at::kFloat;
  DLIA_CHECK(f32 || out.scalar_type() == at::kBFloat16, "out must be bf16 or fp32");
  DLIA_CHECK(epi == 6 || (out.dim() == 2 && out.stride(1) == 1 && out.size(0) >= M && out.size(1) >= n_out), "out shape");
  DLIA_CHECK(out.stride(0) % 4 == 0 && reinterpret_cast<uintptr_t>(out.data_ptr()) % 16 == 0, "out alignment");
  const void* bp = nullptr; const void* rp = nullptr; long ldr = 0;
  if (epi == 1 || epi == 3 || epi == 5) {
    DLIA_CHECK(bias.has_value(), "epilogue needs bias");
    check_bf16(*bias, "bias");
    DLIA_CHECK(bias->is_contiguous() && bias->numel() >= n_out, "bias shape");
    bp = bias->data_ptr();
  }
  if (epi == 2 || epi == 3) {
    DLIA_CHECK(residual.has_value(), "epilogue needs residual");
    check_bf16(*residual, "residual"); check_rows(*residual, "residual");
    DLIA_CHECK(residual->size(0) >= M && residual->size(1) >= n_out, "residual shape");
    rp = residual->data_ptr(); ldr = residual->stride(0);
  }
  auto stream = cur_stream();
  if (M <= 256) {
    float* ws = nullptr;
    int sk = (int)splitk, nt = (int)ntw;
    if (sk <= 0 || nt <= 0) {
      int an, as;
      dlia_gemm_skinny_config((int)M, (int)N, (int)K, (int)epi, workspace.has_value() ? 1 : 0, &an, &as);
      if (sk <= 0) sk = as;
      if (nt <= 0) nt = an;
    }
    if (workspace.has_value()) {
      check_f32(*workspace, "workspace");
      const int64_t have = workspace->numel();
      if (epi == 6) {
        DLIA_CHECK(have >= (int64_t)sk * M * N, "gemm: partial workspace smaller than [splitk, M, N]");
      } else {
        while (sk > 1 && have < (int64_t)sk * M * N) sk /= 2;      // shrink rather than overrun
      }
      ws = workspace->data_ptr<float>();
    } else {
      DLIA_CHECK(epi != 6, "partial output needs a workspace");
      sk = 1;
    }
    dlia_gemm_skinny(out.data_ptr(), x.data_ptr(), wp.data_ptr(), bp, rp, ws, wsc, (int)M, (int)N, (int)K,
                     x.stride(0), out.stride(0), ldr, (int)epi, f32 ? 1 : 0, sk, nt, stream);
  } else if (w8) {
    // prefill over fp8 weights: 256-row slices through the streaming kernel (weights are re-read
    // per slice at half the bytes of bf16; the widening to bf16 happens in registers)
    DLIA_CHECK(epi != 6, "partial output is a decode-regime (M <= 256) feature");
    const int64_t osz = out.element_size();
    for (int64_t m0 = 0; m0 < M; m0 += 256) {
      const int64_t mm = std::min<int64_t>(256, M - m0);
      const char* rps = rp ? (const char*)rp + m0 * ldr * 2 : nullptr;
      dlia_gemm_skinny((char*)out.data_ptr() + m0 * out.stride(0) * osz, (const char*)x.data_ptr() + m0 * x.stride(0) * 2,
                       wp.data_ptr(), bp, rps, nullptr, wsc, (int)mm, (int)N, (int)K, x.stride(0), out.stride(0), ldr,
                       (int)epi, f32 ? 1 : 0, 1, 2, stream);
    }
  } else {
    DLIA_CHECK(epi != 6, "partial output is a decode-regime (M <= 256) feature");
    // A workspace lets the ping-pong kernel split K on shapes with fewer tiles than CUs: slabs in front, and
    // its LAST 4096 words are arrival counters (zero when first handed over, left zero by every launch). A
    // workspace too small for a shape's split simply is not used. For the loader-wave kernel a workspace is
    // only ever the debug buffer of its diagnostic build.
    float* wsp = nullptr; int64_t wsn = 0;
    if (workspace.has_value()) {
      check_f32(*workspace, "workspace");
      DLIA_CHECK(workspace->is_contiguous(), "gemm workspace must be contiguous");
      if (dlia_gemm_large_choice(-1) != 0) {
        const int64_t wgs = ((N + 127) / 128 + 7) / 8 * 8 * ((M + 255) / 256);
        DLIA_CHECK(workspace->numel() >= wgs * 12 * 16, "debug buffer too small");
      }
      wsp = workspace->data_ptr<float>(); wsn = workspace->numel();
    }
    // more than 256 rows: `splitk` > 1 is the number of workgroups the caller wants a thin shape spread over
    // (256: it runs alone; default 128: it shares the chip with another stream)
    dlia_gemm_large(out.data_ptr(), x.data_ptr(), wp.data_ptr(), bp, rp, wsp, wsn, (int)M, (int)N, (int)K,
                    x.stride(0), out.stride(0), ldr, (int)epi, f32 ? 1 : 0, (int)splitk, stream);
  }
}

void gemm(Tensor out, const Tensor& x, const Tensor& wp, OptTensor bias, OptTensor residual,
          OptTensor workspace, int64_t N, int64_t epi, int64_t splitk, int64_t ntw) {
  gemm_impl(out, x, wp, c10::nullopt, bias, residual, workspace, N, epi, splitk, ntw);
}

void gemm_w8(Tensor out, const Tensor& x, const Tensor& wp, const Tensor& wscale, OptTensor bias, OptTensor residual,
             OptTensor workspace, int64_t N, int64_t epi, int64_t splitk, int64_t ntw) {
  gemm_impl(out, x, wp, wscale, bias, residual, workspace, N, epi, splitk, ntw);
}

// q[M,K] (uint8, e4m3), scale[M] (fp32) <- x[M,K] (bf16): per-row dynamic quantisation
void quant_rows(Tensor q, Tensor scale, const Tensor& x) {
  DLIA_ON_DEVICE_OF(q);
  check_bf16(x, "x"); check_rows(x, "x"); check_dev(q, "q"); check_f32(scale, "scale");
  const int64_t M = x.size(0), K = x.size(1);
  DLIA_CHECK(K % 8 == 0, "K must be a multiple of 8");
  DLIA_CHECK(q.scalar_type() == at::kByte && q.dim() == 2 && q.size(0) >= M && q.size(1) >= K && q.stride(1) == 1,
             "q must be uint8 [M, K]");
  DLIA_CHECK(q.stride(0) % 16 == 0 && reinterpret_cast<uintptr_t>(q.data_ptr()) % 16 == 0, "q alignment");
  DLIA_CHECK(scale.is_contiguous() && scale.numel() >= M, "scale shape");
  dlia_quant_rows(q.data_ptr(), scale.data_ptr<float>(), x.data_ptr(), (int)M, (int)K, x.stride(0), q.stride(0), cur_stream());
}

// fp8 x fp8 GEMM: xq uint8 [M,K] + xscale[M], wp from pack_weight_a8w8 + wscale per packed row.
void gemm_a8w8(Tensor out, const Tensor& xq, const Tensor& xscale, const Tensor& wp, const Tensor& wscale,
               OptTensor bias, OptTensor residual, OptTensor workspace, int64_t N, int64_t epi, int64_t splitk,
               int64_t ntw) {
  DLIA_ON_DEVICE_OF(out);
  check_dev(out, "out"); check_dev(xq, "xq"); check_dev(wp, "wp"); check_f32(xscale, "xscale"); check_f32(wscale, "wscale");
  DLIA_CHECK(xq.scalar_type() == at::kByte && xq.dim() == 2 && xq.stride(1) == 1, "xq must be uint8 [M, K]");
  const int64_t M = xq.size(0), K = xq.size(1);
  DLIA_CHECK(K % 128 == 0, "K must be a multiple of 128");
  DLIA_CHECK(xq.stride(0) % 16 == 0 && reinterpret_cast<uintptr_t>(xq.data_ptr()) % 16 == 0, "xq alignment");
  DLIA_CHECK(N % 4 == 0 && epi >= 0 && epi <= 6, "bad N / epilogue");
  const int64_t n_pad = (N + 127) / 128 * 128;
  DLIA_CHECK(wp.is_contiguous() && wp.numel() * wp.element_size() == n_pad * K, "packed weight size");
  DLIA_CHECK(reinterpret_cast<uintptr_t>(wp.data_ptr()) % 16 == 0, "packed weight alignment");
  DLIA_CHECK(wscale.is_contiguous() && wscale.numel() == n_pad, "wscale must hold one scale per packed row");
  DLIA_CHECK(xscale.is_contiguous() && xscale.numel() >= M, "xscale shape");
  const int64_t n_out = epi == 4 ? N / 2 : N;
  if (epi == 4) DLIA_CHECK(N % 32 == 0, "swiglu needs gate/up tile pairs");
  const bool f32 = out.scalar_type() == at::kFloat;
  DLIA_CHECK(f32 || out.scalar_type() == at::kBFloat16, "out must be bf16 or fp32");
  DLIA_CHECK(epi == 6 || (out.dim() == 2 && out.stride(1) == 1 && out.size(0) >= M && out.size(1) >= n_out), "out shape");
  DLIA_CHECK(out.stride(0) % 4 == 0 && reinterpret_cast<uintptr_t>(out.data_ptr()) % 16 == 0, "out alignment");
  const void* bp = nullptr; const void* rp = nullptr; long ldr = 0;
  if (epi == 1 || epi == 3 || epi == 5) {
    DLIA_CHECK(bias.has_value(), "epilogue needs bias");
    check_bf16(*bias, "bias");
    DLIA_CHECK(bias->is_contiguous() && bias->numel() >= n_out, "bias shape");
    bp = bias->data_ptr();
  }
  if (epi == 2 || epi == 3) {
    DLIA_CHECK(residual.has_value(), "epilogue needs residual");
    check_bf16(*residual, "residual"); check_rows(*residual, "residual");
    DLIA_CHECK(residual->size(0) >= M && residual->size(1) >= n_out, "residual shape");
    rp = residual->data_ptr(); ldr = residual->stride(0);
  }
  auto stream = cur_stream();
  if (M <= 256) {
    float* ws = nullptr;
    int sk = (int)splitk;
    if (sk <= 0) {
      int an, as;
      dlia_gemm_skinny_config((int)M, (int)N, (int)K, (int)epi, workspace.has_value() ? 1 : 0, &an, &as);
      sk = as;
    }
    if (workspace.has_value()) {
      check_f32(*workspace, "workspace");
      const int64_t have = workspace->numel();
      if (epi == 6) {
        DLIA_CHECK(have >= (int64_t)sk * M * N, "gemm: partial workspace smaller than [splitk, M, N]");
      } else {
        while (sk > 1 && have < (int64_t)sk * M * N) sk /= 2;
      }
      ws = workspace->data_ptr<float>();
    } else {
      DLIA_CHECK(epi != 6, "partial output needs a workspace");
      sk = 1;
    }
    dlia_gemm_a8w8(out.data_ptr(), xq.data_ptr(), xscale.data_ptr<float>(), wp.data_ptr(), wscale.data_ptr<float>(), bp, rp,
                   ws, (int)M, (int)N, (int)K, xq.stride(0), out.stride(0), ldr, (int)epi, f32 ? 1 : 0, sk, (int)ntw, stream);
  } else {
    DLIA_CHECK(epi != 6, "partial output is a decode-regime (M <= 256) feature");
    // more than 256 rows: one launch of the 256 x 256 ping-pong kernel (or, DLIA_GEMM_LARGE=stream, the loader-wave
    // kernel walking 256-row blocks). A workspace lets thin shapes split K inside the launch, as for bf16: slabs in
    // front, its last 4096 words are arrival counters (zero when first handed over, left zero by every launch).
    float* wsp = nullptr; int64_t wsn = 0;
    if (workspace.has_value()) {
      check_f32(*workspace, "workspace");
      DLIA_CHECK(workspace->is_contiguous(), "gemm workspace must be contiguous");
      wsp = workspace->data_ptr<float>(); wsn = workspace->numel();
    }
    dlia_gemm_a8w8_large(out.data_ptr(), xq.data_ptr(), xscale.data_ptr<float>(), wp.data_ptr(), wscale.data_ptr<float>(), bp, rp,
                         wsp, wsn, (int)M, (int)N, (int)K, xq.stride(0), out.stride(0), ldr, (int)epi, f32 ? 1 : 0, (int)splitk, stream);
  }
}

int64_t gemm_splitk(int64_t M, int64_t N, int64_t K) { return dlia_gemm_skinny_splitk((int)M, (int)N, (int)K); }

// [ntw, splitk] the weight-streaming kernel would use for this shape
std::vector<int64_t> gemm_config(int64_t M, int64_t N, int64_t K, int64_t epi, bool allow_split) {
  int ntw, sk;
  dlia_gemm_skinny_config((int)M, (int)N, (int)K, (int)epi, allow_split ? 1 : 0, &ntw, &sk);
  return {ntw, sk};
}

int64_t attention_ws_floats(int64_t B, int64_t max_q_len, int64_t Hq, int64_t Hkv, int64_t D, int64_t splits) {
  return dlia_paged_attention_ws_floats((int)B, (int)max_q_len, (int)Hq, (int)Hkv, (int)D, (int)splits);
}

static void paged_attention_impl(Tensor out, const Tensor& qkv, const Tensor& kv_cache, OptTensor kv_scale,
                                 const Tensor& q_start, const Tensor& ctx_lens, const Tensor& block_table,
                                 OptTensor slots, OptTensor partial, int64_t max_q_len, int64_t Hq, int64_t Hkv,
                                 int64_t D, double scale, int64_t splits) {
  DLIA_ON_DEVICE_OF(out);
  check_bf16(out, "out"); check_bf16(qkv, "qkv"); check_rows(out, "out"); check_rows(qkv, "qkv");
  DLIA_CHECK(D == 64 || D == 128, "head_dim must be 64 or 128");
  DLIA_CHECK(Hq % Hkv == 0, "Hq must be a multiple of Hkv");
  const int64_t g = Hq / Hkv;
  DLIA_CHECK(g >= 1 && g <= 16, "GQA group size must be between 1 and 16");
  DLIA_CHECK(qkv.size(1) >= Hq * D && out.size(1) >= Hq * D, "qkv/out width");
  auto geo = kv_scale.has_value() ? check_kv8(kv_cache, *kv_scale, Hkv, D) : check_kv(kv_cache, Hkv, D);
  check_i32(q_start, "q_start"); check_i32(ctx_lens, "ctx_lens"); check_i32(block_table, "block_table");
  const int64_t B = ctx_lens.numel();
  DLIA_CHECK(q_start.numel() == B + 1 && block_table.dim() == 2, "attention metadata shapes");
  const int* slp = nullptr;
  if (slots.has_value()) {
    check_i32(*slots, "slots");
    DLIA_CHECK(slots->numel() >= B, "slots shape");
    slp = slots->data_ptr<int>();
  } else {
    DLIA_CHECK(block_table.size(0) >= B, "block_table needs one row per sequence");
  }
  float* pp = nullptr;
  int sp = (int)splits;
  if (sp > 1) {
    DLIA_CHECK(partial.has_value(), "split-KV needs a partial workspace");
    check_f32(*partial, "partial");
    const int64_t need = dlia_paged_attention_ws_floats((int)B, (int)max_q_len, (int)Hq, (int)Hkv, (int)D, sp);
    DLIA_CHECK(partial->numel() >= need, "partial workspace too small: ", partial->numel(), " < ", need);
    pp = partial->data_ptr<float>();
  } else {
    sp = 1;
  }
  // The caller guarantees: q_start is non-decreasing with q_start[B] <= rows of qkv/out, chunk
  // lengths <= max_q_len, and ctx_lens[b] + q_len[b] <= 32 * block_table.size(1).
  dlia_paged_attention(out.data_ptr(), qkv.data_ptr(), kv_cache.data_ptr(),
                       kv_scale.has_value() ? kv_scale->data_ptr<float>() : nullptr, q_start.data_ptr<int>(),
                       ctx_lens.data_ptr<int>(), block_table.data_ptr<int>(), slp, pp, (int)B, (int)max_q_len, (int)Hq,
                       (int)Hkv, (int)D, (int)block_table.size(0), (int)block_table.size(1), (int)geo.num_blocks, qkv.stride(0),
                       out.stride(0), (float)scale, sp, cur_stream());
}

void paged_attention(Tensor out, const Tensor& qkv, const Tensor& kv_cache, const Tensor& q_start,
                     const Tensor& ctx_lens, const Tensor& block_table, OptTensor slots, OptTensor partial,
                     int64_t max_q_len, int64_t Hq, int64_t Hkv, int64_t D, double scale, int64_t splits) {
  paged_attention_impl(out, qkv, kv_cache, c10::nullopt, q_start, ctx_lens, block_table, slots, partial, max_q_len, Hq, Hkv, D, scale, splits);
}

// The same over the e4m3 cache.
void paged_attention8(Tensor out, const Tensor& qkv, const Tensor& kv_cache, const Tensor& kv_scale, const Tensor& q_start,
                      const Tensor& ctx_lens, const Tensor& block_table, OptTensor slots, OptTensor partial,
                      int64_t max_q_len, int64_t Hq, int64_t Hkv, int64_t D, double scale, int64_t splits) {
  paged_attention_impl(out, qkv, kv_cache, kv_scale, q_start, ctx_lens, block_table, slots, partial, max_q_len, Hq, Hkv, D, scale, splits);
}

static void sample_impl(Tensor out_tokens, const Tensor& logits, const Tensor& temperature, const Tensor& top_k,
                        const Tensor& top_p, const Tensor& seeds, const Tensor& steps, int64_t vocab,
                        OptTensor dbg_probs, OptTensor dbg_ids, OptTensor chunk_max, OptTensor ids_out = c10::nullopt,
                        bool advance_steps = false, OptTensor ctx_advance = c10::nullopt) {
  DLIA_ON_DEVICE_OF(out_tokens);
  check_i32(out_tokens, "out_tokens"); check_dev(logits, "logits");
  DLIA_CHECK(logits.scalar_type() == at::kFloat && logits.dim() == 2 && logits.stride(1) == 1, "logits must be fp32 [R, >=vocab]");
  const int64_t R = logits.size(0);
  DLIA_CHECK(vocab > 0 && vocab <= logits.size(1), "vocab out of range");
  DLIA_CHECK(logits.stride(0) % 4 == 0 && reinterpret_cast<uintptr_t>(logits.data_ptr()) % 16 == 0,
             "logits rows must be 16-byte aligned");
  check_f32(temperature, "temperature"); check_i32(top_k, "top_k"); check_f32(top_p, "top_p"); check_i32(steps, "steps");
  check_dev(seeds, "seeds");
  DLIA_CHECK(seeds.scalar_type() == at::kLong && seeds.is_contiguous(), "seeds must be int64");
  DLIA_CHECK(out_tokens.numel() >= R && temperature.numel() >= R && top_k.numel() >= R && top_p.numel() >= R &&
             seeds.numel() >= R && steps.numel() >= R, "per-row sampling parameter shapes");
  float* dp = nullptr; int* di = nullptr; int cap = 0;
  if (dbg_probs.has_value()) {
    DLIA_CHECK(dbg_ids.has_value(), "dbg_probs needs dbg_ids");
    check_f32(*dbg_probs, "dbg_probs"); check_i32(*dbg_ids, "dbg_ids");
    DLIA_CHECK(dbg_probs->dim() == 2 && dbg_probs->size(0) >= R && dbg_ids->sizes() == dbg_probs->sizes(), "dbg shapes");
    dp = dbg_probs->data_ptr<float>(); di = dbg_ids->data_ptr<int>(); cap = (int)dbg_probs->size(1);
    DLIA_CHECK(cap <= 1024, "dbg capacity <= 1024");
  }
  const float* cm = nullptr; long ldc = 0;
  if (chunk_max.has_value()) {
    // one maximum per 32-column chunk of each row, as gemm_head writes them
    check_dev(*chunk_max, "chunk_max");
    DLIA_CHECK(chunk_max->scalar_type() == at::kFloat && chunk_max->dim() == 2 && chunk_max->stride(1) == 1 &&
               chunk_max->size(0) >= R && chunk_max->size(1) >= (vocab + 31) / 32, "chunk_max must be fp32 [R, ceil(vocab / 32)]");
    cm = chunk_max->data_ptr<float>(); ldc = chunk_max->stride(0);
  }
  int* io = nullptr; int* ca = nullptr;
  if (ids_out.has_value()) {
    check_i32(*ids_out, "ids_out");
    DLIA_CHECK(ids_out->numel() >= R && ids_out->is_contiguous(), "ids_out shape");
    io = ids_out->data_ptr<int>();
  }
  if (ctx_advance.has_value()) {
    check_i32(*ctx_advance, "ctx");
    DLIA_CHECK(ctx_advance->numel() == R && ctx_advance->is_contiguous(), "ctx must hold one context length per sampled row");
    ca = ctx_advance->data_ptr<int>();
  }
  if (advance_steps) DLIA_CHECK(steps.numel() == R && steps.is_contiguous(), "steps must hold one counter per sampled row");
  dlia_sample(out_tokens.data_ptr<int>(), logits.data_ptr<float>(), temperature.data_ptr<float>(),
              top_k.data_ptr<int>(), top_p.data_ptr<float>(), reinterpret_cast<const uint64_t*>(seeds.data_ptr<int64_t>()),
              steps.data_ptr<int>(), (int)R, (int)vocab, logits.stride(0), dp, di, cap, cur_stream(), cm, ldc, io,
              advance_steps ? steps.data_ptr<int>() : nullptr, ca);
}

void sample(Tensor out_tokens, const Tensor& logits, const Tensor& temperature, const Tensor& top_k,
            const Tensor& top_p, const Tensor& seeds, const Tensor& steps, int64_t vocab,
            OptTensor dbg_probs, OptTensor dbg_ids) {
  sample_impl(out_tokens, logits, temperature, top_k, top_p, seeds, steps, vocab, dbg_probs, dbg_ids, c10::nullopt);
}

// The same with the chunk maxima of `gemm_head`: the kernel reads the row only where a candidate can be.
void sample_guided(Tensor out_tokens, const Tensor& logits, const Tensor& chunk_max, const Tensor& temperature,
                   const Tensor& top_k, const Tensor& top_p, const Tensor& seeds, const Tensor& steps, int64_t vocab,
                   OptTensor dbg_probs, OptTensor dbg_ids) {
  sample_impl(out_tokens, logits, temperature, top_k, top_p, seeds, steps, vocab, dbg_probs, dbg_ids, chunk_max);
}

// The sampler of a captured decode step: the draw of `sample` / `sample_guided` (chunk_max given), and by the thread that
// writes a row's token also the step's bookkeeping — the token into `ids_out` (the next step's input ids), the row's
// draw counter in `steps` and its context length in `ctx` advanced by one — instead of three more launches.
void sample_step(Tensor out_tokens, const Tensor& logits, OptTensor chunk_max, const Tensor& temperature, const Tensor& top_k,
                 const Tensor& top_p, const Tensor& seeds, Tensor steps, int64_t vocab, OptTensor ids_out, OptTensor ctx) {
  sample_impl(out_tokens, logits, temperature, top_k, top_p, seeds, steps, vocab, c10::nullopt, c10::nullopt, chunk_max, ids_out,
              true, ctx);
}

// lm_head at any row count (up to 256: the weight-streaming kernels, above: the tile kernel): logits[M, N] (fp32) = x[M, K] @ Wp^T, and chunk_max[M, ceil(N / 32)] = the maximum of
// every 32-column chunk of every row over the columns < vocab (the sampler's guide). `workspace` as for gemm.
void gemm_head(Tensor logits, Tensor chunk_max, const Tensor& x, const Tensor& wp, OptTensor workspace, int64_t N,
               int64_t vocab) {
  DLIA_ON_DEVICE_OF(logits);
  check_bf16(x, "x"); check_rows(x, "x"); check_dev(logits, "logits"); check_dev(wp, "wp"); check_dev(chunk_max, "chunk_max");
  const int64_t M = x.size(0), K = x.size(1);
  DLIA_CHECK(M >= 1, "gemm_head: no rows");
  DLIA_CHECK(K % 128 == 0 && N % 4 == 0 && vocab > 0 && vocab <= N, "gemm_head: K % 128, N % 4, 0 < vocab <= N");
  const int64_t n_pad = (N + 127) / 128 * 128;
  DLIA_CHECK(wp.is_contiguous() && wp.numel() * wp.element_size() == n_pad * K * 2, "packed weight size");
  DLIA_CHECK(reinterpret_cast<uintptr_t>(wp.data_ptr()) % 16 == 0, "packed weight alignment");
  DLIA_CHECK(logits.scalar_type() == at::kFloat && logits.dim() == 2 && logits.stride(1) == 1 && logits.size(0) >= M &&
             logits.size(1) >= N && logits.stride(0) % 4 == 0 && reinterpret_cast<uintptr_t>(logits.data_ptr()) % 16 == 0,
             "logits must be fp32 [M, >= N], 16-byte aligned rows");
  // the kernels store a maximum for every chunk of every 256-column (tile kernel) / 128-column block they compute
  const int64_t chunks = (n_pad + 255) / 256 * 8;
  DLIA_CHECK(chunk_max.scalar_type() == at::kFloat && chunk_max.dim() == 2 && chunk_max.stride(1) == 1 &&
             chunk_max.size(0) >= M && chunk_max.size(1) >= chunks, "chunk_max must be fp32 [M, >= ", chunks, "]");
  if (M <= 256) {
    dlia_gemm_skinny_head(logits.data_ptr(), x.data_ptr(), wp.data_ptr(), (int)M, (int)N, (int)K, x.stride(0), logits.stride(0),
                          chunk_max.data_ptr<float>(), chunk_max.stride(0), (int)vocab, cur_stream());
    return;
  }
  float* wsp = nullptr; int64_t wsn = 0;
  if (workspace.has_value()) {
    check_f32(*workspace, "workspace");
    DLIA_CHECK(workspace->is_contiguous(), "gemm workspace must be contiguous");
    wsp = workspace->data_ptr<float>(); wsn = workspace->numel();
  }
  dlia_gemm_large(logits.data_ptr(), x.data_ptr(), wp.data_ptr(), nullptr, nullptr, wsp, wsn, (int)M, (int)N, (int)K,
                  x.stride(0), logits.stride(0), 0, 0, 1, 0, cur_stream(), chunk_max.data_ptr<float>(), chunk_max.stride(0),
                  (int)vocab);
}

// dst[rows[r], :] = src[r, :] (int32 table rows; rows outside dst are skipped)
void scatter_rows(Tensor dst, const Tensor& rows, const Tensor& src) {
  DLIA_ON_DEVICE_OF(dst);
  check_i32(dst, "dst"); check_i32(rows, "rows"); check_i32(src, "src");
  DLIA_CHECK(dst.dim() == 2 && src.dim() == 2 && src.size(1) == dst.size(1) && rows.dim() == 1 && rows.numel() == src.size(0),
             "scatter_rows: dst [R, C], rows [n], src [n, C]");
  dlia_scatter_rows_i32(dst.data_ptr<int>(), rows.data_ptr<int>(), src.data_ptr<int>(), (int)rows.numel(), (int)dst.size(1),
                        (int)dst.size(0), cur_stream());
}

// dst.flat[idx[i]] = val[i] (int32; indices outside dst are skipped)
void scatter(Tensor dst, const Tensor& idx, const Tensor& val) {
  DLIA_ON_DEVICE_OF(dst);
  check_i32(dst, "dst"); check_i32(idx, "idx"); check_i32(val, "val");
  DLIA_CHECK(idx.dim() == 1 && val.dim() == 1 && idx.numel() == val.numel(), "scatter: idx [n], val [n]");
  dlia_scatter_i32(dst.data_ptr<int>(), idx.data_ptr<int>(), val.data_ptr<int>(), (int)idx.numel(), (long)dst.numel(), cur_stream());
}

void advance(Tensor x, int64_t delta) {
  DLIA_ON_DEVICE_OF(x);
  check_i32(x, "x");
  dlia_advance_i32(x.data_ptr<int>(), (int)x.numel(), (int)delta, cur_stream());
}

}  // namespace

TORCH_LIBRARY(dlia, m) {
  m.def("norm(Tensor(a!) out, Tensor(b!) x, Tensor? delta, Tensor? partial, int splits, Tensor? pbias, Tensor w, Tensor? b, Tensor? row_idx, float eps, int mode) -> ()");
  m.def("embed(Tensor(a!) out, Tensor ids, Tensor? pos, Tensor wte, Tensor? wpe) -> ()");
  m.def("rope_kvwrite(Tensor(a!) qkv, Tensor? partial, int splits, Tensor? pbias, Tensor(b!) kv_cache, Tensor cos, Tensor sin, Tensor tok_pos, Tensor tok_seq, Tensor block_table, int Hq, int Hkv, int D, bool use_rope) -> ()");
  m.def("rope_kvwrite8(Tensor(a!) qkv, Tensor? partial, int splits, Tensor? pbias, Tensor(b!) kv_cache, Tensor(c!) kv_scale, Tensor cos, Tensor sin, Tensor tok_pos, Tensor tok_seq, Tensor block_table, int Hq, int Hkv, int D, bool use_rope) -> ()");
  m.def("paged_attention8(Tensor(a!) out, Tensor qkv, Tensor kv_cache, Tensor kv_scale, Tensor q_start, Tensor ctx_lens, Tensor block_table, Tensor? slots, Tensor? partial, int max_q_len, int Hq, int Hkv, int D, float scale, int splits) -> ()");
  m.def("gemm(Tensor(a!) out, Tensor x, Tensor wp, Tensor? bias, Tensor? residual, Tensor? workspace, int N, int epi, int splitk, int ntw) -> ()");
  m.def("gemm_w8(Tensor(a!) out, Tensor x, Tensor wp, Tensor wscale, Tensor? bias, Tensor? residual, Tensor? workspace, int N, int epi, int splitk, int ntw) -> ()");
  m.def("gemm_a8w8(Tensor(a!) out, Tensor xq, Tensor xscale, Tensor wp, Tensor wscale, Tensor? bias, Tensor? residual, Tensor? workspace, int N, int epi, int splitk, int ntw) -> ()");
  m.def("norm_q(Tensor(a!) out_q, Tensor(b!) qscale, Tensor(c!) x, Tensor? delta, Tensor? partial, int splits, Tensor? pbias, Tensor w, Tensor? b, float eps, int mode) -> ()");
  m.def("quant_rows(Tensor(a!) q, Tensor(b!) scale, Tensor x) -> ()");
  m.def("gemm_config(int M, int N, int K, int epi, bool allow_split) -> int[]");
  m.def("gemm_splitk(int M, int N, int K) -> int");
  m.def("large_gemm_kernel(int which) -> int");
  m.def("gemm_ws_floats(int M, int N, int K) -> int");
  m.def("attention_ws_floats(int B, int max_q_len, int Hq, int Hkv, int D, int splits) -> int");
  m.def("paged_attention(Tensor(a!) out, Tensor qkv, Tensor kv_cache, Tensor q_start, Tensor ctx_lens, Tensor block_table, Tensor? slots, Tensor? partial, int max_q_len, int Hq, int Hkv, int D, float scale, int splits) -> ()");
  m.def("sample(Tensor(a!) out_tokens, Tensor logits, Tensor temperature, Tensor top_k, Tensor top_p, Tensor seeds, Tensor steps, int vocab, Tensor? dbg_probs, Tensor? dbg_ids) -> ()");
  m.def("sample_step(Tensor(a!) out_tokens, Tensor logits, Tensor? chunk_max, Tensor temperature, Tensor top_k, Tensor top_p, Tensor seeds, Tensor(b!) steps, int vocab, Tensor(c!)? ids_out, Tensor(d!)? ctx) -> ()");
  m.def("sample_guided(Tensor(a!) out_tokens, Tensor logits, Tensor chunk_max, Tensor temperature, Tensor top_k, Tensor top_p, Tensor seeds, Tensor steps, int vocab, Tensor? dbg_probs, Tensor? dbg_ids) -> ()");
  m.def("gemm_head(Tensor(a!) logits, Tensor(b!) chunk_max, Tensor x, Tensor wp, Tensor? workspace, int N, int vocab) -> ()");
  m.def("advance(Tensor(a!) x, int delta) -> ()");
  m.def("scatter_rows(Tensor(a!) dst, Tensor rows, Tensor src) -> ()");
  m.def("scatter(Tensor(a!) dst, Tensor idx, Tensor val) -> ()");
}

TORCH_LIBRARY_IMPL(dlia, CUDA, m) {
  m.impl("norm", &norm);
  m.impl("embed", &embed);
  m.impl("rope_kvwrite", &rope_kvwrite);
  m.impl("rope_kvwrite8", &rope_kvwrite8);
  m.impl("paged_attention8", &paged_attention8);
  m.impl("gemm", &gemm);
  m.impl("gemm_w8", &gemm_w8);
  m.impl("gemm_a8w8", &gemm_a8w8);
  m.impl("quant_rows", &quant_rows);
  m.impl("norm_q", &norm_q);
  m.impl("paged_attention", &paged_attention);
  m.impl("sample", &sample);
  m.impl("sample_guided", &sample_guided);
  m.impl("sample_step", &sample_step);
  m.impl("gemm_head", &gemm_head);
  m.impl("advance", &advance);
  m.impl("scatter_rows", &scatter_rows);
  m.impl("scatter", &scatter);
}

TORCH_LIBRARY_IMPL(dlia, CompositeExplicitAutograd, m) {
  m.impl("gemm_splitk", &gemm_splitk);
  m.impl("large_gemm_kernel", &large_gemm_kernel);
  m.impl("gemm_ws_floats", &gemm_ws_floats);
  m.impl("gemm_config", &gemm_config);
  m.impl("attention_ws_floats", &attention_ws_floats);
}
