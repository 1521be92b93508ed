// Token embedding gather (+ learned position embedding for GPT-2). Stage 0 only.
// Replaces `embed_tokens(generated_ids)` which the reference runs over the WHOLE sequence on
// every step (orchestration.py:111); here only the new tokens are gathered.
#include "common.h"
#include "launch.h"

namespace dlia {

__global__ __launch_bounds__(128) void embed_kernel(
    bf16* __restrict__ out, const int* __restrict__ ids, const int* __restrict__ pos,
    const bf16* __restrict__ wte, const bf16* __restrict__ wpe, int H, int vocab, int max_pos) {
  const int t = blockIdx.x;
  int id = ids[t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // never fault on a bad id
  const bf16* src = wte + (long)id * H;
  bf16* dst = out + (long)t * H;
  const int nvec = H >> 3;
  if (wpe) {
    int p = pos[t];
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    const bf16* ps = wpe + (long)p * H;
    for (int i = threadIdx.x; i < nvec; i += 128) {
      float a[8], b[8];
      unpack8(ld16(src + i * 8), a);
      unpack8(ld16(ps + i * 8), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += b[j];
      st16(dst + i * 8, pack8(a));
    }
  } else {
    for (int i = threadIdx.x; i < nvec; i += 128) st16(dst + i * 8, ld16(src + i * 8));
  }
}

__global__ void advance_kernel(int* x, int n, int delta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += delta;
}

// Block-table maintenance from a step's staged metadata (one pinned upload per step, no per-slot
// host copies): whole rows for sequences whose table is new to the stage, single entries for
// sequences that grew by a block. Indices out of range are skipped, never written.
__global__ void scatter_rows_kernel(int* __restrict__ dst, const int* __restrict__ rows, const int* __restrict__ src,
                                    int nrows, int ncols, int dst_rows) {
  const int r = blockIdx.x;
  const int row = rows[r];
  if (row < 0 || row >= dst_rows) return;
  for (int c = threadIdx.x; c < ncols; c += blockDim.x) dst[(long)row * ncols + c] = src[(long)r * ncols + c];
}

__global__ void scatter_kernel(int* __restrict__ dst, const int* __restrict__ idx, const int* __restrict__ val, int n, long limit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = idx[i];
  if (j >= 0 && j < limit) dst[j] = val[i];
}

}  // namespace dlia

extern "C" void dlia_scatter_rows_i32(int* dst, const int* rows, const int* src, int nrows, int ncols, int dst_rows,
                                      hipStream_t stream) {
  if (nrows <= 0 || ncols <= 0) return;
  hipLaunchKernelGGL(dlia::scatter_rows_kernel, dim3(nrows), dim3(64), 0, stream, dst, rows, src, nrows, ncols, dst_rows);
}

extern "C" void dlia_scatter_i32(int* dst, const int* idx, const int* val, int n, long limit, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(dlia::scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dst, idx, val, n, limit);
}

extern "C" void dlia_embed(void* out, const int* ids, const int* pos, const void* wte, const void* wpe,
                           int T, int H, int vocab, int max_pos, hipStream_t stream) {
  if (T <= 0) return;
  hipLaunchKernelGGL(dlia::embed_kernel, dim3(T), dim3(128), 0, stream, (dlia::bf16*)out, ids, pos,
                     (const dlia::bf16*)wte, (const dlia::bf16*)wpe, H, vocab, max_pos);
}

extern "C" void dlia_advance_i32(int* x, int n, int delta, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(dlia::advance_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, n, delta);
}
