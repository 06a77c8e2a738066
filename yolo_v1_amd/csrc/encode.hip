// Target encoder on the device: utils/YOLODataLoader.py:200-230 for a whole batch.
// Compiled with -ffp-contract=off: cell index and in-cell offset follow the reference's fp32 op sequence
// (divide by fp32(1/S), ceil, subtract, divide) so the targets are bit-identical to the host encoder.
#include "common.h"
#include "yv1.h"

// One thread per (image, cell).  The reference writes boxes in order and a later box landing in an occupied cell
// replaces it completely, so the cell's content is that of the LAST box that maps to it -- found by a scan over the
// image's (few) boxes, which needs no inter-thread ordering.  Python's negative indexing is kept: a coordinate of
// exactly 0 gives index -1, which the reference's `target[int(ij[1]), int(ij[0])]` wraps to S-1.
__global__ void __launch_bounds__(256) k_encode_targets(const float* __restrict__ boxes, const long long* __restrict__ labels,
                                                        const int* __restrict__ counts, int N, int Kmax, int S, int B, int C,
                                                        float cell, float* __restrict__ target, int* __restrict__ err) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * S * S) return;
  const int n = idx / (S * S), cellid = idx - n * S * S;
  const int row = cellid / S, col = cellid - row * S;
  const int k_img = min(counts[n], Kmax);
  int last = -1;
  float dx = 0.f, dy = 0.f;
  for (int k = 0; k < k_img; ++k) {
    const float* b = boxes + ((size_t)n * Kmax + k) * 4;
    const float cx = b[0], cy = b[1];
    const float fi = ceilf(cx / cell) - 1.0f, fj = ceilf(cy / cell) - 1.0f;
    int ci = (int)fi, cj = (int)fj;
    if (ci < 0) ci += S;
    if (cj < 0) cj += S;
    const long long lab = labels[(size_t)n * Kmax + k];
    if (ci < 0 || ci >= S || cj < 0 || cj >= S || lab < 0 || lab >= C) {   // the reference raises IndexError here
      if (cellid == 0) atomicOr(err, 1);
      continue;
    }
    if (ci == col && cj == row) {
      last = k;
      dx = (cx - fi * cell) / cell;
      dy = (cy - fj * cell) / cell;
    }
  }
  const int D = B * 5 + C;
  float* t = target + (size_t)idx * D;
  if (last < 0) {
    for (int d = 0; d < D; ++d) t[d] = 0.f;
    return;
  }
  const float* b = boxes + ((size_t)n * Kmax + last) * 4;
  const float w = b[2], h = b[3];
  const int lab = (int)labels[(size_t)n * Kmax + last];
  for (int d = 0; d < B; ++d) t[d] = 1.f;
  for (int s = 0; s < B; ++s) {
    t[B + s * 4 + 0] = dx;
    t[B + s * 4 + 1] = dy;
    t[B + s * 4 + 2] = w;
    t[B + s * 4 + 3] = h;
  }
  for (int c = 0; c < C; ++c) t[B * 5 + c] = (c == lab) ? 1.f : 0.f;
}

extern "C" int yv1_encode_targets(const float* boxes, const long long* labels, const int* counts, int N, int Kmax, int S,
                                  int B, int C, float* target, int* err_flag, yv1_stream_t stream) {
  if (N < 0 || Kmax < 0 || S <= 0 || S > 64 || B <= 0 || C <= 0 || !target || !counts || !err_flag) return YV1_ERR_BAD_ARG;
  if (Kmax > 0 && (!boxes || !labels)) return YV1_ERR_BAD_ARG;
  if (N == 0) return YV1_OK;
  hipStream_t st = (hipStream_t)stream;
  YV1_HIP(hipMemsetAsync(err_flag, 0, sizeof(int), st));
  const int total = N * S * S;
  const float cell = (float)(1.0 / (double)S);   // `cell_size = 1./self.S` (a Python double) meets the fp32 tensor as fp32
  hipLaunchKernelGGL(k_encode_targets, dim3(cdiv(total, 256)), dim3(256), 0, st, boxes, labels, counts, N, Kmax, S, B, C,
                     cell, target, err_flag);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}
