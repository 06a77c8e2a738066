// Fused multi-tensor SGD with momentum for gfx950.
//
// Replaces torch.optim.SGD(lr, momentum=0.99).step() of the reference (train.py:84, :172), which
// runs ~5 foreach kernels over 161 tensors (read w,g,m / write m,w several times).  One launch
// handles up to YV1_SGD_MAX_TENSORS tensors: buf = momentum*buf + grad ; w -= lr*buf, in one pass
// (16 B per lane, 12 B read + 8 B written per element -- HBM-bound, 41 M parameters = 0.8 GB/step).
// The learning rate is read from device memory, so the per-iteration LR schedule (train.py:158-160)
// does not change any kernel argument and the whole training step can be replayed from a hipGraph.
// torch semantics kept: dampening 0, no nesterov, no weight decay; the momentum buffer starts at
// zero, which makes the first step buf = grad exactly as torch initialises it.
#include "common.h"

#define YV1_SGD_MAX_TENSORS 48

namespace {

struct SgdTable {
  float* w[YV1_SGD_MAX_TENSORS];
  const float* g[YV1_SGD_MAX_TENSORS];
  float* m[YV1_SGD_MAX_TENSORS];
  long long n[YV1_SGD_MAX_TENSORS];
  int first_block[YV1_SGD_MAX_TENSORS + 1];   // prefix sum of per-tensor block counts
  unsigned char vec[YV1_SGD_MAX_TENSORS];     // all three pointers 16-byte aligned -> float4 path
  int count;
};

constexpr int SGD_ELEMS_PER_BLOCK = 256 * 4 * 4;   // 256 threads x float4 x 4 iterations

__global__ void __launch_bounds__(256) k_sgd(SgdTable t, const float* __restrict__ lr_ptr, float momentum, float grad_scale) {
  // locate the tensor this block belongs to (count <= 48: linear scan in SGPRs)
  int ti = 0;
  while (ti + 1 < t.count && (int)blockIdx.x >= t.first_block[ti + 1]) ++ti;
  const long long base = (long long)(blockIdx.x - t.first_block[ti]) * SGD_ELEMS_PER_BLOCK;
  const long long n = t.n[ti];
  float* __restrict__ w = t.w[ti];
  const float* __restrict__ g = t.g[ti];
  float* __restrict__ m = t.m[ti];
  const float lr = *lr_ptr;
  const bool vec = t.vec[ti] != 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long long i = base + (long long)(it * 256 + threadIdx.x) * 4;
    if (vec && i + 3 < n) {
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 mv = *reinterpret_cast<const float4*>(m + i);
      float4 wv = *reinterpret_cast<const float4*>(w + i);
      mv.x = momentum * mv.x + gv.x * grad_scale; mv.y = momentum * mv.y + gv.y * grad_scale;
      mv.z = momentum * mv.z + gv.z * grad_scale; mv.w = momentum * mv.w + gv.w * grad_scale;
      wv.x -= lr * mv.x; wv.y -= lr * mv.y; wv.z -= lr * mv.z; wv.w -= lr * mv.w;
      *reinterpret_cast<float4*>(m + i) = mv;
      *reinterpret_cast<float4*>(w + i) = wv;
    } else {
      for (long long j = i; j < n && j < i + 4; ++j) {
        const float mv = momentum * m[j] + g[j] * grad_scale;
        m[j] = mv;
        w[j] -= lr * mv;
      }
    }
  }
}

}  // namespace

extern "C" int yv1_sgd_max_tensors(void) { return YV1_SGD_MAX_TENSORS; }

// One fused step over `count` (<= yv1_sgd_max_tensors()) dense fp32 tensors given as HOST arrays of
// device pointers; w/g/m of a tensor share one memory order (16-byte aligned tensors take the float4 path).
// *lr is read on the device.  grad_scale multiplies the gradient (1/world_size after an all-reduce sum).
extern "C" int yv1_sgd_momentum_step(float* const* w, const float* const* g, float* const* m, const long long* n, int count,
                                     const float* lr, float momentum, float grad_scale, hipStream_t stream) {
  if (!w || !g || !m || !n || !lr || count <= 0 || count > YV1_SGD_MAX_TENSORS) return YV1_ERR_BAD_ARG;
  SgdTable t;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (!w[i] || !g[i] || !m[i] || n[i] <= 0) return YV1_ERR_BAD_ARG;
    t.vec[i] = ((((uintptr_t)w[i] | (uintptr_t)g[i] | (uintptr_t)m[i]) & 15) == 0) ? 1 : 0;
    t.w[i] = w[i]; t.g[i] = g[i]; t.m[i] = m[i]; t.n[i] = n[i];
    t.first_block[i] = blocks;
    blocks += (int)((n[i] + SGD_ELEMS_PER_BLOCK - 1) / SGD_ELEMS_PER_BLOCK);
  }
  t.first_block[count] = blocks;
  t.count = count;
  hipLaunchKernelGGL(k_sgd, dim3(blocks), dim3(256), 0, stream, t, lr, momentum, grad_scale);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}
