// "bn3's backward as algebra" (DESIGN.md section 7): the small matrix kernels.
//
// An identity-shortcut Bottleneck ends  out = relu(bn3(conv3(z2)) + x)  (backbones/OriginResNet.py:97-105 of the reference).
// conv3 is pointwise (y3 = z2 W3^T), so BatchNorm-3's backward commutes with it.  With gm the ReLU-masked block-output
// gradient [M][4p], T = gm^T z2 (the weight-gradient GEMM, fed gm), G = z2^T z2, the column sums sum(gm), sum(z2), the
// batch statistics (mu, is) of y3 and  k1 = gamma*is, k2 = k1*dbeta/M, k3 = k1*dgamma/M:
//     dbeta = sum(gm)                       dgamma[c] = is[c] * (W3[c,:] . T[c,:] - mu[c]*dbeta[c])      (no pass over y3)
//     dz2   = gm (diag(k1) W3) - z2 (W3^T diag(k3*is) W3) + sum_c (k3*is*mu - k2)[c] W3[c,:]
//     dW3   = diag(k1) T - diag(k3*is) (W3 G - mu (x) sum(z2)) - k2 (x) sum(z2)
// The stand-alone reduce and apply passes over the 4p-wide tensors (gm, y3 -> dy3) disappear; dy3 is never formed.
// W3 here is the bf16 copy the forward convolution multiplied with ([4p][p], K contiguous), so the identities hold for the
// y3 the network actually produced.  Everything is fp32, fixed summation order (bitwise reproducible).
#include "common.h"

namespace {

// one workgroup per channel c: column sum of the partial rows, the row dot W3[c,:] . T[c,:], then the coefficients
__global__ void __launch_bounds__(256) k_bn3_coeffs(const float* __restrict__ gsum, int rows, const float* __restrict__ T,
                                                    const bf16_t* __restrict__ W3, int p, int C4, const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, const float* __restrict__ gamma, float count,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ k1,
                                                    float* __restrict__ k2, float* __restrict__ k3is) {
  __shared__ float red[2][256];
  const int c = blockIdx.x, tid = threadIdx.x;
  float s = 0.f, d = 0.f;
  for (int r = tid; r < rows; r += 256) s += gsum[(size_t)r * C4 + c];
  for (int j = tid; j < p; j += 256) d += bf16_to_f32(W3[(size_t)c * p + j]) * T[(size_t)c * p + j];
  red[0][tid] = s; red[1][tid] = d;
  __syncthreads();
  if (tid == 0) {
    float ss = 0.f, dd = 0.f;
    for (int i = 0; i < 256; ++i) { ss += red[0][i]; dd += red[1][i]; }
    const float is = invstd[c], mu = mean[c];
    const float dg = is * (dd - mu * ss);
    const float a1 = gamma[c] * is;
    dbeta[c] = ss;
    dgamma[c] = dg;
    k1[c] = a1;
    k2[c] = a1 * ss / count;
    k3is[c] = a1 * dg / count * is;
  }
}

// wcat [p][5p] bf16 (row j = conv3 input channel = dgrad output channel):  wcat[j][c] = k1[c] W3[c][j]  (c < 4p),
// wcat[j][4p + i] = -Q[i][j],  Q = W3^T diag(k3is) W3 (symmetric);  bias[j] = sum_c (k3is*mu - k2)[c] W3[c][j].
// grid (p/32, p/32 + C4/32 + 1): blockIdx.y < p/32: a 32x32 tile of Q; < p/32 + C4/32: a tile of the scaled copy; last: bias
__global__ void __launch_bounds__(256) k_bn3_build(const bf16_t* __restrict__ W3, int p, int C4, const float* __restrict__ k1,
                                                   const float* __restrict__ k2, const float* __restrict__ k3is,
                                                   const float* __restrict__ mean, bf16_t* __restrict__ wcat,
                                                   float* __restrict__ bias) {
  __shared__ float sa[32][33], sb[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  const int j0 = blockIdx.x * 32;
  const int K5 = C4 + p;
  const int qt = p / 32, ct = C4 / 32;
  if ((int)blockIdx.y < qt) {
    // Q[i][j] for i in [i0, i0+32), j in [j0, j0+32): sum over c of W3[c][i] * k3is[c] * W3[c][j]
    const int i0 = blockIdx.y * 32;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < C4; c0 += 32) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + 8 * r;
        sa[ty + 8 * r][tx] = bf16_to_f32(W3[(size_t)c * p + i0 + tx]) * k3is[c];
        sb[ty + 8 * r][tx] = bf16_to_f32(W3[(size_t)c * p + j0 + tx]);
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = ty + 8 * r;
        float t = acc[r];
#pragma unroll 8
        for (int cc = 0; cc < 32; ++cc) t += sa[cc][il] * sb[cc][tx];
        acc[r] = t;
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + ty + 8 * r, j = j0 + tx;
      wcat[(size_t)j * K5 + C4 + i] = f32_to_bf16(-acc[r]);
    }
  } else if ((int)blockIdx.y < qt + ct) {
    const int c0 = ((int)blockIdx.y - qt) * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = c0 + ty + 8 * r;
      sa[ty + 8 * r][tx] = bf16_to_f32(W3[(size_t)c * p + j0 + tx]) * k1[c];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + ty + 8 * r;
      wcat[(size_t)j * K5 + c0 + tx] = f32_to_bf16(sa[tx][ty + 8 * r]);
    }
  } else {
    // bias[j0 + tx]: 8 row lanes, fixed combine order
    float t = 0.f;
    for (int c = ty; c < C4; c += 8) t += (k3is[c] * mean[c] - k2[c]) * bf16_to_f32(W3[(size_t)c * p + j0 + tx]);
    sa[ty][tx] = t;
    __syncthreads();
    if (ty == 0) {
      float b = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) b += sa[r][tx];
      bias[j0 + tx] = b;
    }
  }
}

// dW3[c][j] = k1[c] T[c][j] - k3is[c] (sum_i W3[c][i] G[i][j] - mu[c] sz[j]) - k2[c] sz[j];  sz = column sum of the partial
// rows szp [rows][2][p] (first half: sum of z2).  One 32x32 tile per workgroup.
__global__ void __launch_bounds__(256) k_bn3_dw(const float* __restrict__ T, const float* __restrict__ G,
                                                const float* __restrict__ szp, int rows, const bf16_t* __restrict__ W3, int p,
                                                int C4, const float* __restrict__ k1, const float* __restrict__ k2,
                                                const float* __restrict__ k3is, const float* __restrict__ mean,
                                                float* __restrict__ dW) {
  __shared__ float sa[32][33], sb[32][33], ssz[32], sred[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  {
    // column sums of the partial rows: eight row lanes per column, two loads in flight each, combined in a fixed order (one
    // serial chain per column over ~1000 rows took 200 us per launch: the whole kernel was this loop)
    float t0 = 0.f, t1 = 0.f;
    int r = ty;
    for (; r + 8 < rows; r += 16) {
      t0 += szp[(size_t)r * 2 * p + j0 + tx];
      t1 += szp[(size_t)(r + 8) * 2 * p + j0 + tx];
    }
    if (r < rows) t0 += szp[(size_t)r * 2 * p + j0 + tx];
    sred[ty][tx] = t0 + t1;
    __syncthreads();
    if (ty == 0) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) t += sred[i][tx];
      ssz[tx] = t;
    }
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i0 = 0; i0 < p; i0 += 32) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sa[ty + 8 * r][tx] = bf16_to_f32(W3[(size_t)(c0 + ty + 8 * r) * p + i0 + tx]);      // [c][i]
      sb[ty + 8 * r][tx] = G[(size_t)(i0 + ty + 8 * r) * p + j0 + tx];                     // [i][j]
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cl = ty + 8 * r;
      float t = acc[r];
#pragma unroll 8
      for (int ii = 0; ii < 32; ++ii) t += sa[cl][ii] * sb[ii][tx];
      acc[r] = t;
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + ty + 8 * r, j = j0 + tx;
    const float sz = ssz[tx];
    dW[(size_t)c * p + j] = k1[c] * T[(size_t)c * p + j] - k3is[c] * (acc[r] - mean[c] * sz) - k2[c] * sz;
  }
}

}  // namespace

// gsum: [rows][C4] partial column sums of gm (yv1_conv2d_dgrad_add_masked_out_nhwc_bf16); T: fp32 [C4][p] = gm^T z2
// (yv1_conv2d_wgrad_nhwc_bf16 with dy := gm); w3: the forward's bf16 weights [C4][p]; count = pixels M.
// Writes BatchNorm-3's parameter gradients and the three coefficient vectors [C4] each.
extern "C" int yv1_bn3_coeffs(const float* gsum, int rows, const float* T, const void* w3, int p, int C4, const float* mean,
                              const float* invstd, const float* gamma, float count, float* dgamma, float* dbeta, float* k1,
                              float* k2, float* k3is, hipStream_t stream) {
  if (!gsum || rows <= 0 || !T || !w3 || p <= 0 || C4 <= 0 || !mean || !invstd || !gamma || !dgamma || !dbeta || !k1 || !k2 || !k3is)
    return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_bn3_coeffs, dim3(C4), dim3(256), 0, stream, gsum, rows, T, (const bf16_t*)w3, p, C4, mean, invstd, gamma,
                     count, dgamma, dbeta, k1, k2, k3is);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// wcat: bf16 [p][C4 + p], bias: fp32 [p] -- the operands of yv1_conv2d_dgrad_cat_bias_nhwc_bf16
extern "C" int yv1_bn3_build(const void* w3, int p, int C4, const float* k1, const float* k2, const float* k3is,
                             const float* mean, void* wcat, float* bias, hipStream_t stream) {
  if (!w3 || !k1 || !k2 || !k3is || !mean || !wcat || !bias) return YV1_ERR_BAD_ARG;
  if (p % 32 || C4 % 32) return YV1_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_bn3_build, dim3(p / 32, p / 32 + C4 / 32 + 1), dim3(256), 0, stream, (const bf16_t*)w3, p, C4, k1, k2, k3is,
                     mean, (bf16_t*)wcat, bias);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// dW: fp32 [C4][p] (conv3's weight gradient, the layout yv1_conv2d_wgrad_nhwc_bf16 writes); G: fp32 [p][p] = z2^T z2;
// sz_partials: [rows][2][p] from yv1_bn_stats(z2)
extern "C" int yv1_bn3_dw(const float* T, const float* G, const float* sz_partials, int rows, const void* w3, int p, int C4,
                          const float* k1, const float* k2, const float* k3is, const float* mean, float* dW,
                          hipStream_t stream) {
  if (!T || !G || !sz_partials || rows <= 0 || !w3 || !k1 || !k2 || !k3is || !mean || !dW) return YV1_ERR_BAD_ARG;
  if (p % 32 || C4 % 32) return YV1_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_bn3_dw, dim3(p / 32, C4 / 32), dim3(256), 0, stream, T, G, sz_partials, rows, (const bf16_t*)w3, p, C4, k1,
                     k2, k3is, mean, dW);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}
