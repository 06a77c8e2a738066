// Deferred BatchNorm backward: the per-channel part.
//
// relu(bn(x)) feeds a pointwise convolution (DenseNet's norm1 -> relu1 -> conv1 over the concatenated features,
// OriginDenseNet.py:22-27,:32-36, and the transitions' norm -> relu -> conv, :50-52).  With d the gradient at the BatchNorm
// output (masked by the ReLU), a = gamma*invstd, xhat = (x - mean)*invstd and M pixels, the BatchNorm backward is
//     dx = a*d  -  a*mean(d)  -  a*xhat*mean(d*xhat)
// The first term needs no reduction: the data gradient's epilogue adds it to the gradient buffer directly
// (yv1_conv2d_dgrad_bn_deferred_nhwc_bf16, conv.hip).  The other two are AFFINE in x per channel,
//     corr(x) = A + B*x,   B = a*dgamma*invstd/M,   A = a*dbeta/M - B*mean          (dbeta = sum d, dgamma = sum d*xhat)
// and the features x of a dense block are the same tensor for every layer that normalises them: the coefficients of all
// those layers are summed per channel (KA, KB) and subtracted ONCE, just before the gradient of a channel is consumed
// (yv1_bn_deferred_fix).  fp32, fixed summation order.
#include "common.h"

namespace {

// block = 32 channels x 32 row groups; partial rows [rows][2][C]
__global__ void __launch_bounds__(1024) k_bn_bwd_finalize_deferred(const float* __restrict__ part, int rows, int C, float count,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, float* __restrict__ KA,
                                                                   float* __restrict__ KB, int accumulate) {
  __shared__ float red[2][32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int r = ty; r < rows; r += 32) {
      s1 += part[((size_t)r * 2 + 0) * C + c];
      s2 += part[((size_t)r * 2 + 1) * C + c];
    }
  red[0][ty][tx] = s1;
  red[1][ty][tx] = s2;
  __syncthreads();
  if (ty == 0 && c < C) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { t1 += red[0][i][tx]; t2 += red[1][i][tx]; }
    const float is = invstd[c], mu = mean[c];
    const float a = gamma[c] * is;
    const float dg = is * t2;                       // sum d * xhat
    dbeta[c] = t1;
    dgamma[c] = dg;
    const float B = a * dg * is / count;
    const float A = a * t1 / count - B * mu;
    KA[c] = accumulate ? KA[c] + A : A;
    KB[c] = accumulate ? KB[c] + B : B;
  }
}

// g[m][c] -= KA[c] + KB[c] * x[m][c] over a C-channel window (C % 8 == 0) of two NHWC bf16 tensors.
// FIXED: the grid stride is a multiple of the chunks per pixel, so a thread keeps its 8 channels -- their coefficients are
// loaded once, not per element (they were 64 B of loads beside 32 B of data).
template <bool FIXED>
__global__ void __launch_bounds__(256) k_bn_deferred_fix(bf16_t* __restrict__ g, int ldg, const bf16_t* __restrict__ x, int ldx,
                                                         const float* __restrict__ KA, const float* __restrict__ KB,
                                                         long long npix, int C) {
  const int cpr = C >> 3;
  const long long total = npix * cpr;
  const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  float ka[8], kb[8];
  if (FIXED) {
    const int c8 = (int)(i0 % cpr) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) { ka[k] = KA[c8 + k]; kb[k] = KB[c8 + k]; }
  }
  for (long long i = i0; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / cpr;
    const int c8 = (int)(i - m * cpr) * 8;
    if (!FIXED) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { ka[k] = KA[c8 + k]; kb[k] = KB[c8 + k]; }
    }
    const uint4 gv = *reinterpret_cast<const uint4*>(g + m * ldg + c8);
    const uint4 xv = *reinterpret_cast<const uint4*>(x + m * ldx + c8);
    const unsigned* pg = reinterpret_cast<const unsigned*>(&gv);
    const unsigned* px = reinterpret_cast<const unsigned*>(&xv);
    unsigned res[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = __uint_as_float(pg[k] << 16) - (ka[2 * k] + kb[2 * k] * __uint_as_float(px[k] << 16));
      const float hi = __uint_as_float(pg[k] & 0xffff0000u) - (ka[2 * k + 1] + kb[2 * k + 1] * __uint_as_float(px[k] & 0xffff0000u));
      res[k] = pack_bf16x2(lo, hi);
    }
    *reinterpret_cast<uint4*>(g + m * ldg + c8) = make_uint4(res[0], res[1], res[2], res[3]);
  }
}

}  // namespace

// part: [rows][2][C] partial sums of yv1_conv2d_dgrad_bn_deferred_nhwc_bf16 (rows <= 2048: pre-reduce longer tables with
// yv1_reduce_rows); gamma / mean / invstd: the BatchNorm's weight and batch statistics; writes dgamma, dbeta [C] and adds
// (accumulate) or stores the correction coefficients into KA, KB [C].
extern "C" int yv1_bn_bwd_finalize_deferred(const float* part, int rows, int C, float count, const float* gamma, const float* mean,
                                            const float* invstd, float* dgamma, float* dbeta, float* KA, float* KB,
                                            int accumulate, hipStream_t stream) {
  if (!part || !gamma || !mean || !invstd || !dgamma || !dbeta || !KA || !KB || rows <= 0 || C <= 0) return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_bn_bwd_finalize_deferred, dim3((C + 31) / 32), dim3(1024), 0, stream, part, rows, C, count, gamma, mean,
                     invstd, dgamma, dbeta, KA, KB, accumulate);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// g, x: channel windows (C channels, C % 8 == 0, 16-byte aligned rows) of NHWC bf16 tensors with pixel strides ldg / ldx
extern "C" int yv1_bn_deferred_fix(void* g, int ldg, const void* x, int ldx, const float* KA, const float* KB, long long npix,
                                   int C, hipStream_t stream) {
  if (!g || !x || !KA || !KB || npix <= 0 || C <= 0) return YV1_ERR_BAD_ARG;
  if (C % 8 || ldg % 8 || ldx % 8) return YV1_ERR_UNSUPPORTED;
  const long long total = npix * (C / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if ((blocks * 256) % (C / 8) == 0)
    hipLaunchKernelGGL(k_bn_deferred_fix<true>, dim3((unsigned)blocks), dim3(256), 0, stream, (bf16_t*)g, ldg, (const bf16_t*)x,
                       ldx, KA, KB, npix, C);
  else
    hipLaunchKernelGGL(k_bn_deferred_fix<false>, dim3((unsigned)blocks), dim3(256), 0, stream, (bf16_t*)g, ldg, (const bf16_t*)x,
                       ldx, KA, KB, npix, C);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}
