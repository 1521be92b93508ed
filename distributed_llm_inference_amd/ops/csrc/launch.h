// C launch entry points of the HIP kernel library. The kernels are compiled without any
// PyTorch headers; bindings.cpp wraps these in torch custom ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" {

// ---- norm.hip -------------------------------------------------------------------------
// out[r,:] = norm(x[row_idx ? row_idx[r] : r,:] (+ addend, written back into x)) * w (+ b).
// addend: `delta` (bf16 rows) or the fp32 split-K partials [splits][pM][H] of the preceding GEMM
// (+ pbias). mode: 0 RMSNorm, 1 LayerNorm, 2 only fold the addend into x.
void dlia_norm(void* out, void* x, const void* delta, const float* partial, int splits, long pM,
               const void* pbias, const void* w, const void* b, const int* row_idx, int rows, int H,
               long ldx, long ldo, float eps, int mode, float* qscale, hipStream_t stream);
// qscale != null: `out` receives e4m3 bytes (row stride ldo BYTES) and qscale[r] the row's scale

// ---- embed.hip ------------------------------------------------------------------------
// out[t,:] = wte[ids[t],:] (+ wpe[pos[t],:])
void dlia_embed(void* out, const int* ids, const int* pos, const void* wte, const void* wpe,
                int T, int H, int vocab, int max_pos, hipStream_t stream);

// ---- rope_kv.hip ----------------------------------------------------------------------
// RoPE on the q and k slices of a fused qkv activation (in place on q), and scatter of the
// (rotated) k and v rows into the paged, MFMA-fragment-ordered KV cache.
// The q/k/v values come from `qkv` itself, or (partial != null) from the fp32 split-K partials
// [splits][T][(Hq+2Hkv)*D] (+ pbias) of the QKV GEMM; the rotated q is written to `qkv` either way.
void dlia_rope_kvwrite(void* qkv, const float* partial, int splits, const void* pbias, void* kv_cache,
                       float* kv_scale, const float* cos_t, const float* sin_t,
                       const int* tok_pos, const int* tok_seq, const int* block_table,
                       int T, int Hq, int Hkv, int D, int max_blocks, int num_blocks,
                       int rope_len, int use_rope, hipStream_t stream);

// ---- gemm_skinny.hip ------------------------------------------------------------------
// out[M,N] = epilogue(x[M,K] @ Wp^T), M <= 128 per call (weight-streaming regime).
// epi: 0 none, 1 +bias, 2 +residual, 3 +bias+residual, 4 swiglu (interleaved gate/up tiles,
// N = 2*N_out), 5 +bias then gelu_tanh, 6 fp32 partials [splitk][M][N] only (summed by the
// consumer kernel). out_f32: write fp32 instead of bf16.
void dlia_gemm_skinny_head(void* logits, const void* x, const void* wp, int M, int N, int K, long ldx, long ldo,
                           float* chunk_max, long ldc, int n_valid, hipStream_t stream);
void dlia_gemm_skinny(void* out, const void* x, const void* wp, const void* bias,
                      const void* residual, float* workspace, const float* wscale,
                      int M, int N, int K, long ldx, long ldo, long ldr, int epi, int out_f32,
                      int splitk, int ntw, hipStream_t stream);
// the decomposition the launcher would choose: n-tiles per wave and split-K factor
void dlia_gemm_skinny_config(int M, int N, int K, int epi, int allow_split, int* ntw_out, int* splitk_out);
int dlia_gemm_skinny_splitk(int M, int N, int K);

// ---- gemm_a8w8.hip --------------------------------------------------------------------
// fp8 x fp8 streaming GEMM on the block-scaled MFMA (M <= 256): xq = e4m3 bytes [M,K] with per-row
// scales, wp = pack_weight_a8w8 image with per-column scales. Epilogues as dlia_gemm_skinny.
void dlia_gemm_a8w8(void* out, const void* xq, const float* xscale, const void* wp, const float* wscale,
                    const void* bias, const void* residual, float* workspace, int M, int N, int K,
                    long ldx, long ldo, long ldr, int epi, int out_f32, int splitk, int ntw, hipStream_t stream);
// per-row dynamic e4m3 quantisation of bf16 activations: q[M,K] bytes (row stride ldq), scale[M]
void dlia_quant_rows(void* q, float* scale, const void* x, int M, int K, long ldx, long ldq, hipStream_t stream);

// ---- gemm_rows.hip / gemm_pingpong.hip (more than 128 / 256 rows; reached through dlia_gemm_skinny) ----
// 1: the streaming GEMM picks 128- / 64-row blocks on shapes with few column blocks (off by default); returns the old value
int dlia_gemm_rows_set_narrow(int on);
// bf16 GEMM of more than 256 rows, same epilogues as dlia_gemm_skinny (0..5). workspace / ws_floats (optional):
// room for the in-launch split-K of thin shapes; its LAST 4096 words are arrival counters, zero between launches.
// chunk_max (optional, the ping-pong kernel only): per row the maximum of every 32-column chunk of the output
// (columns >= n_valid excluded), chunk_max[m * ldc + n / 32].
void dlia_gemm_large(void* out, const void* x, const void* wp, const void* bias, const void* residual,
                     float* workspace, long ws_floats, int M, int N, int K, long ldx, long ldo, long ldr,
                     int epi, int out_f32, int wg_target, hipStream_t stream, float* chunk_max = nullptr, long ldc = 0,
                     int n_valid = 0);
extern "C" void dlia_gemm_a8w8_large(void* out, const void* xq, const float* xscale, const void* wp, const float* wscale,
                                     const void* bias, const void* residual, float* workspace, long ws_floats, int M,
                                     int N, int K, long ldx, long ldo, long ldr, int epi, int out_f32, int wg_target,
                                     hipStream_t stream);
// kernel for bf16 GEMMs of more than 256 rows: 0 = 256 x 256 ping-pong tiles (default), 2 = the loader-wave
// kernel of gemm_rows.hip; which < 0 only reads; returns the previous choice
int dlia_gemm_large_choice(int which);
// the large-M kernel's in-launch split-K: factor for a shape, and the workspace floats it needs (slabs + counters)
int dlia_gemm_pingpong_splitk(int M, int N, int K);
long dlia_gemm_pingpong_ws_floats(int M, int N, int K);

// ---- attention.hip --------------------------------------------------------------------
// Causal GQA attention of ragged query chunks against the paged KV cache (prefill chunks and
// single-token decode are the same kernel). kv_scale != null: e4m3 cache + per-(token, head) scales. Work items are (sequence, kv head, query tile,
// kv split); `partial` holds split-KV partials when splits > 1.
void dlia_paged_attention(void* out, const void* qkv, const void* kv_cache, const float* kv_scale,
                          const int* q_start, const int* ctx_lens, const int* block_table,
                          const int* slots, float* partial, int B, int max_q_len, int Hq,
                          int Hkv, int D, int table_rows, int max_blocks, int num_blocks,
                          long ldq, long ldo, float scale, int splits, hipStream_t stream);
long dlia_paged_attention_ws_floats(int B, int max_q_len, int Hq, int Hkv, int D, int splits);

// ---- sampler.hip ----------------------------------------------------------------------
// One token per row from fp32 logits [R, ldl] with temperature / top-k / top-p and a
// counter-based RNG keyed by (seed[r], step[r]); temperature <= 0 -> argmax.
// chunk_max (optional): per row the maximum of every 32-column chunk of its logits (row stride ldc, at most
// 8192 chunks), as the lm_head GEMM writes them; the row is then read only where a candidate can be.
void dlia_sample(int* out_tokens, const float* logits, const float* temperature,
                 const int* top_k, const float* top_p, const uint64_t* seeds,
                 const int* steps, int R, int vocab, long ldl, float* dbg_probs,
                 int* dbg_ids, int dbg_cap, hipStream_t stream, const float* chunk_max = nullptr, long ldc = 0,
                 int* ids_out = nullptr, int* steps_adv = nullptr, int* ctx_adv = nullptr);

// ---- misc.hip -------------------------------------------------------------------------
void dlia_advance_i32(int* x, int n, int delta, hipStream_t stream);
void dlia_scatter_rows_i32(int* dst, const int* rows, const int* src, int nrows, int ncols, int dst_rows, hipStream_t stream);
void dlia_scatter_i32(int* dst, const int* idx, const int* val, int n, long limit, hipStream_t stream);

}  // extern "C"
