// BatchNorm (training mode) forward/backward, pooling, the sigmoid head and weight re-layout for
// gfx950.  All HBM-bound: every kernel moves 16 B (8 bf16 channels) per lane along the contiguous
// NHWC channel axis, keeps per-channel vectors in registers and reduces over pixels with
// column-parallel accumulators (no atomics; partial rows are summed in a fixed order).
//
// Replaces the ATen/cuDNN kernels behind nn.BatchNorm2d / nn.ReLU / residual add
// (backbones/OriginResNet.py:90-105, :174-177, :187; backbones/OriginDenseNet.py:22-27, :50-51,
// :125), nn.MaxPool2d(3,2,1) (OriginResNet.py:125, OriginDenseNet.py:80), nn.AvgPool2d(2,2)
// (OriginDenseNet.py:54) and torch.sigmoid + permute (OriginResNet.py:188-189).
//
// Training-mode BN is split so that no extra pass over the activation is spent on statistics:
//   conv epilogue  -> per-tile partial (sum, sumsq)            (csrc/conv.hip)
//   yv1_bn_finalize-> mean, invstd, scale=g*invstd, shift=b-mean*scale, running stats
//   yv1_bn_apply   -> z = relu?(scale*y + shift [+ residual | + rscale*res + rshift])
// Backward:  yv1_bn_bwd_reduce (sum dyh, sum dyh*xhat, relu mask recomputed from z or from
// scale*y+shift) -> yv1_bn_bwd_finalize (dgamma, dbeta, coefficients) -> yv1_bn_bwd_apply.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(v[k] << 16);
    f[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = pack_bf16x2(f[2 * k], f[2 * k + 1]);
  return v;
}
__device__ __forceinline__ void load8f(const float* p, float* f) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// ------------------------------------------------------------------ row reduction of partials
// in [rows][W] -> out [gridDim.y][W], each block-row sums RB consecutive rows (fixed order).
__global__ void k_reduce_rows(const float* __restrict__ in, float* __restrict__ out, int rows, int W, int RB) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= W) return;
  const int r0 = blockIdx.y * RB, r1 = min(rows, r0 + RB);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += in[(size_t)r * W + c]; s1 += in[(size_t)(r + 1) * W + c];
    s2 += in[(size_t)(r + 2) * W + c]; s3 += in[(size_t)(r + 3) * W + c];
  }
  for (; r < r1; ++r) s0 += in[(size_t)r * W + c];
  out[(size_t)blockIdx.y * W + c] = (s0 + s1) + (s2 + s3);
}

// partial (sum, sumsq) rows [rows][2][Cseg] -> one row inside a wider [2][ldo] table at channel c0.
// 1024 threads = 64 columns x 16 row lanes (rows can be thousands: one serial chain per column took 100 us).
__global__ void __launch_bounds__(1024) k_stats_merge(const float* __restrict__ in, int rows, int Cseg, float* __restrict__ out,
                                                      int ldo, int c0) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  float s0 = 0.f, s1 = 0.f;
  if (i < 2 * Cseg) {
    int r = ty;
    for (; r + 16 < rows; r += 32) {
      s0 += in[(size_t)r * 2 * Cseg + i];
      s1 += in[(size_t)(r + 16) * 2 * Cseg + i];
    }
    for (; r < rows; r += 16) s0 += in[(size_t)r * 2 * Cseg + i];
  }
  red[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty != 0 || i >= 2 * Cseg) return;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += red[k][tx];
  const int half = i / Cseg, c = i - half * Cseg;
  out[(size_t)half * ldo + c0 + c] = s;
}

// ------------------------------------------------------------------ BN forward finalize
// 1024 threads: tx = channel (64 per workgroup), ty = row lane (16): every row lane sums rows ty, ty+4, ... with 4 independent loads in flight, the 4 partial sums are combined
// through LDS in a fixed order, then lane 0 does the per-channel math.  One launch replaces the former
// reduce_rows + finalize pair.
template <int CPB>
__global__ void __launch_bounds__(1024) k_bn_finalize(const float* part, int rows, int C, int ldp, float count,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float momentum, float* running_mean, float* running_var,
                                                     float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                     float* __restrict__ scale_out, float* __restrict__ shift_out,
                                                     const float* __restrict__ seg_part, int seg_rows, int seg_c0, int seg_c,
                                                     float* table_out) {
  // CPB channels per workgroup x RL = 1024/CPB row lanes.  CPB 64 reads full 256-B lines; CPB 16 spreads a narrow
  // BatchNorm (C <= 1024) over 4x the workgroups -- these launches are latency-bound, not bandwidth-bound.
  constexpr int RL = 1024 / CPB;
  __shared__ float red[2][RL][CPB];
  const int tx = threadIdx.x % CPB, ty = threadIdx.x / CPB;
  const int c = blockIdx.x * CPB + tx;
  float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
  // merged form (DenseNet: yv1_bn_finalize_merged): channels [seg_c0, seg_c0 + seg_c) are not in the table yet -- their
  // sums are still the partial rows [seg_rows][2][seg_c] of the convolution that produced them; they are reduced here
  // and written into the table for the BatchNorms further down the block
  const bool seg = seg_part != nullptr && c >= seg_c0 && c < seg_c0 + seg_c;
  if (c < C) {
    const float* src = seg ? seg_part + (c - seg_c0) : part + c;
    const size_t rstride = seg ? (size_t)2 * seg_c : (size_t)2 * ldp, qoff = seg ? seg_c : ldp;
    const int nrows = seg ? seg_rows : rows;
    int r = ty;
    for (; r + RL < nrows; r += 2 * RL) {
      s0 += src[(size_t)r * rstride];
      q0 += src[(size_t)r * rstride + qoff];
      s1 += src[(size_t)(r + RL) * rstride];
      q1 += src[(size_t)(r + RL) * rstride + qoff];
    }
    for (; r < nrows; r += RL) {
      s0 += src[(size_t)r * rstride];
      q0 += src[(size_t)r * rstride + qoff];
    }
  }
  red[0][ty][tx] = s0 + s1;
  red[1][ty][tx] = q0 + q1;
  __syncthreads();
  if (ty != 0 || c >= C) return;
  float s = 0.f, ss = 0.f;
#pragma unroll 16
  for (int j = 0; j < RL; ++j) { s += red[0][j][tx]; ss += red[1][j][tx]; }
  if (seg) { table_out[c] = s; table_out[ldp + c] = ss; }
  const float mean = s / count;
  float var = ss / count - mean * mean;
  var = var < 0.f ? 0.f : var;
  const float invstd = rsqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean_out[c] = mean;
  invstd_out[c] = invstd;
  scale_out[c] = g * invstd;
  shift_out[c] = b - mean * g * invstd;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    const float unb = count > 1.f ? var * count / (count - 1.f) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
  }
}

// eval mode: scale/shift from running statistics
__global__ void k_bn_eval_coeffs(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                 float* __restrict__ scale_out, float* __restrict__ shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = rsqrtf(rv[c] + eps);
  scale_out[c] = gamma[c] * invstd;
  shift_out[c] = beta[c] - rm[c] * gamma[c] * invstd;
}

// ------------------------------------------------------------------ BN apply (+ReLU, +residual)
struct ApplyArgs {
  const bf16_t* y; int ldy;
  bf16_t* z; int ldz;
  const bf16_t* res; int ldr;     // residual (nullable)
  const float* scale; const float* shift;
  const float* rscale; const float* rshift;   // when set, residual is a raw conv output with its own BN
  long long npix; int C; int relu;
  unsigned char* relu_mask;       // optional [npix][C/8]: bit k of byte j = output channel 8j+k is > 0
  unsigned char* z8 = nullptr;    // optional second output: e4m3 of the bf16 result (operand of the next fp8 convolution)
  int ldz8 = 0;
};

// FIXED_C: the grid stride (256 * gridDim.x) is a multiple of CH (the host rounds the grid: fixed_chunk_grid), so a thread
// keeps its 8-channel chunk for the whole loop and the per-channel vectors are loaded once instead of once per 16 bytes of
// data -- they were 4 (8 with a BatchNorm'ed residual) of the 6 (11) vector-memory instructions of an iteration.
template <bool FIXED_C>
__global__ void __launch_bounds__(256) k_bn_apply(ApplyArgs a) {
  const int CH = a.C >> 3;
  const long long total = a.npix * CH;
  float sc[8], sh[8], rs[8], rh[8];
  if (FIXED_C) {
    const int cc = (int)((blockIdx.x * 256u + threadIdx.x) % CH) * 8;
    load8f(a.scale + cc, sc);
    load8f(a.shift + cc, sh);
    if (a.res && a.rscale) { load8f(a.rscale + cc, rs); load8f(a.rshift + cc, rh); }
  }
  // FIXED_C: the pixel index advances by a constant (grid stride / CH) -- no 64-bit division per 16 bytes of data
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x;
  const long long dpix = FIXED_C ? (long long)gridDim.x * 256 / CH : 0;
  long long pix_f = i0 / (unsigned)CH;
  const int cc_f = (int)(i0 % (unsigned)CH) * 8;
  for (long long i = i0; FIXED_C ? pix_f < a.npix : i < total; i += (long long)gridDim.x * 256, pix_f += dpix) {
    const long long pix = FIXED_C ? pix_f : i / CH;
    const int cc = FIXED_C ? cc_f : (int)(i - pix * CH) * 8;
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.y + pix * a.ldy + cc), f);
    if (!FIXED_C) {
      load8f(a.scale + cc, sc);
      load8f(a.shift + cc, sh);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = f[k] * sc[k] + sh[k];
    if (a.res) {
      float r[8];
      unpack8(*reinterpret_cast<const u32x4*>(a.res + pix * a.ldr + cc), r);
      if (a.rscale) {
        if (!FIXED_C) {
          load8f(a.rscale + cc, rs);
          load8f(a.rshift + cc, rh);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = r[k] * rs[k] + rh[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] += r[k];
    }
    if (a.relu) {
      if (a.relu_mask) {
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) m |= (f[k] > 0.f ? 1u : 0u) << k;
        a.relu_mask[pix * CH + (cc >> 3)] = (unsigned char)m;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k], 0.f);
    }
    const u32x4 zb = pack8(f);
    *reinterpret_cast<u32x4*>(a.z + pix * a.ldz + cc) = zb;
    if (a.z8) {                              // quantise the STORED (bf16-rounded) values, as the fp8 inference path does
      float r[8];
      unpack8(zb, r);
      uint2 o;
      o.x = pack_e4m3x4(r[0], r[1], r[2], r[3]);
      o.y = pack_e4m3x4(r[4], r[5], r[6], r[7]);
      *reinterpret_cast<uint2*>(a.z8 + pix * a.ldz8 + cc) = o;
    }
  }
}

// ------------------------------------------------------------------ column-parallel pixel reductions
// Thread layout: TX threads across 16-B channel chunks (TX = pow2 <= CH), TY = 256/TX pixel lanes;
// a thread owns chunk columns tx and tx+TX (CH < 2*TX).  Per-block partial rows go to
// out[blockIdx.x][NV][C].
template <int NV, int NCOL>
__device__ __forceinline__ void block_column_reduce(float (&acc)[NCOL][NV][8], int tx, int ty, int TX, int TY, int CH, int C,
                                                    float* __restrict__ out_row) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [TY][NV][C]
#pragma unroll
  for (int j = 0; j < NCOL; ++j) {
    const int col = tx + j * TX;
    if (col < CH) {
      // two 16-byte stores per vector: as eight dword stores (lane stride 32 B) these were 8-way bank conflicts, 71 % of the
      // kernels' LDS-active cycles (profiles/r02_pmc_sq_summary.txt); C % 8 == 0 keeps them aligned
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float4* dst = reinterpret_cast<float4*>(&red[((size_t)ty * NV + v) * C + col * 8]);
        dst[0] = make_float4(acc[j][v][0], acc[j][v][1], acc[j][v][2], acc[j][v][3]);
        dst[1] = make_float4(acc[j][v][4], acc[j][v][5], acc[j][v][6], acc[j][v][7]);
      }
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < NV * C; idx += 256) {
    float s = 0.f;
    for (int t = 0; t < TY; ++t) s += red[(size_t)t * NV * C + idx];
    out_row[idx] = s;
  }
}

struct StatsArgs {
  const bf16_t* y; int ldy; long long npix; int C; int pix_per_block; float* part;  // [blocks][2][C]
};

template <int NCOL>
__global__ void __launch_bounds__(256) k_bn_stats(StatsArgs a, int TX) {
  const int CH = a.C >> 3, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  float acc[NCOL][2][8];
#pragma unroll
  for (int j = 0; j < NCOL; ++j)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[j][v][k] = 0.f;
  const long long p0 = (long long)blockIdx.x * a.pix_per_block;
  const long long p1 = min(a.npix, p0 + a.pix_per_block);
#pragma unroll 2
  for (long long p = p0 + ty; p < p1; p += TY) {
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
      const int col = tx + j * TX;
      if (col < CH) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(a.y + p * a.ldy + col * 8), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc[j][0][k] += f[k]; acc[j][1][k] += f[k] * f[k]; }
      }
    }
  }
  block_column_reduce<2, NCOL>(acc, tx, ty, TX, TY, CH, a.C, a.part + (size_t)blockIdx.x * 2 * a.C);
}

struct BwdArgs {
  const bf16_t* dz; int lddz;      // grad wrt BN(+ReLU) output
  const bf16_t* z; int ldz;        // output (ReLU mask), used when mask_mode == 1
  const bf16_t* y; int ldy;        // BN input (raw conv output)
  const float* mean; const float* invstd; const float* scale; const float* shift;
  long long npix; int C; int pix_per_block;
  int mask_mode;                   // 0 none, 1 z > 0, 2 scale*y+shift > 0, 3 z is a bit mask [npix][ldz bytes]
  float* part;                     // reduce: [blocks][2][C]  (sum dyh, sum dyh*xhat)
  // apply:
  const float* k1; const float* k2; const float* k3;   // dy = k1*dyh - k2 - xhat*k3 ... see finalize
  bf16_t* dy; int lddy;
  bf16_t* dres; int lddres;        // optional: masked dz copied out (identity shortcut gradient)
  int accumulate;                  // dy += result (DenseNet: several consumers of one feature map)
  // dual form (projection Bottleneck, OriginResNet.py:100-105: out = relu(bn3(y3) + bn_d(yd))): a second BatchNorm whose
  // output gradient is the SAME masked dz -- one pass reads dz and the mask once for both
  const bf16_t* y2 = nullptr; int ldy2 = 0;
  const float* mean2 = nullptr; const float* invstd2 = nullptr;
  float* part2 = nullptr;
  const float* k1b = nullptr; const float* k2b = nullptr; const float* k3b = nullptr;
  bf16_t* dy2 = nullptr; int lddy2 = 0;
  // pooled form (stem): dz is the gradient of a 3x3/2 max pool's OUTPUT [N,OH,OW,C]; pool_idx the pool's first-argmax
  // codes; the pixel index p runs over the pool's INPUT [N,pH,pW], whose gradient is gathered on the fly
  const unsigned char* pool_idx = nullptr; int pH = 0, pW = 0;
};

// gradient of the max pool's input pixel p, channels c8..c8+7: the (up to four) windows that contain the pixel hand
// their gradient over when their stored argmax code names it (same arithmetic and summation order as k_maxpool_bwd_idx)
__device__ __forceinline__ void pooled_grad(const BwdArgs& a, long long p, int c8, float* g) {
  const unsigned H = a.pH, W = a.pW, up = (unsigned)p;
  const unsigned w = up % W, t = up / W, h = t % H, n = t / H;
  const int OH = (int)(H + 2 - 3) / 2 + 1, OW = (int)(W + 2 - 3) / 2 + 1;
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] = 0.f;
  // an input pixel lies in one pooled window per axis when its coordinate is even, in two when it is odd: the (up to) four
  // windows are visited as a fixed 2x2 with predicates -- all index / gradient loads are issued before the first compare
  // (the loop form with run-time bounds made four dependent round trips to L2 per thread: 1.8 TB/s for the whole kernel)
  const int oh0 = h / 2, ow0 = w / 2;
  const bool vh1 = (h & 1u) && oh0 + 1 <= OH - 1, vw1 = (w & 1u) && ow0 + 1 <= OW - 1;
  uint2 am2[2][2];
  u32x4 dv[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oh = oh0 + ((i && vh1) ? 1 : 0), ow = ow0 + ((j && vw1) ? 1 : 0);      // clamped: always a valid address
      const size_t o = (size_t)(n * OH + oh) * OW + ow;
      am2[i][j] = *reinterpret_cast<const uint2*>(a.pool_idx + o * a.C + c8);
      dv[i][j] = *reinterpret_cast<const u32x4*>(a.dz + o * a.lddz + c8);
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool valid = (i == 0 || vh1) && (j == 0 || vw1);
      const int oh = oh0 + i, ow = ow0 + j;
      const unsigned code = (unsigned)(((int)h - (oh * 2 - 1)) * 3 + ((int)w - (ow * 2 - 1)));
      float d[8];
      unpack8(dv[i][j], d);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned am = ((k < 4 ? am2[i][j].x : am2[i][j].y) >> (8 * (k & 3))) & 0xffu;
        g[k] += (valid && am == code) ? d[k] : 0.f;
      }
    }
  // the unfused path stores this gradient as bf16 before the BatchNorm backward reads it: round the same way
  unpack8(pack8(g), g);
}

// sc/sh: scale/shift of the channel chunk when the caller already holds them (mask_mode 2), else nullptr
__device__ __forceinline__ void masked_grad(const BwdArgs& a, long long p, int c8, const float* yv, float* g,
                                            const float* sc_in = nullptr, const float* sh_in = nullptr) {
  if (a.pool_idx) pooled_grad(a, p, c8, g);
  else unpack8(*reinterpret_cast<const u32x4*>(a.dz + p * a.lddz + c8), g);
  if (a.mask_mode == 1) {
    float zv[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.z + p * a.ldz + c8), zv);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = zv[k] > 0.f ? g[k] : 0.f;
  } else if (a.mask_mode == 3) {
    const unsigned m = reinterpret_cast<const unsigned char*>(a.z)[p * a.ldz + (c8 >> 3)];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = ((m >> k) & 1u) ? g[k] : 0.f;
  } else if (a.mask_mode == 2) {
    float sc[8], sh[8];
    if (sc_in) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { sc[k] = sc_in[k]; sh[k] = sh_in[k]; }
    } else {
      load8f(a.scale + c8, sc);
      load8f(a.shift + c8, sh);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = (yv[k] * sc[k] + sh[k]) > 0.f ? g[k] : 0.f;
  }
}

template <int NCOL>
__global__ void __launch_bounds__(256) k_bn_bwd_reduce(BwdArgs a, int TX) {
  const int CH = a.C >> 3, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  float acc[NCOL][2][8];
  float mu[NCOL][8], is[NCOL][8], sc[NCOL][8], sh[NCOL][8];
#pragma unroll
  for (int j = 0; j < NCOL; ++j) {
    const int col = tx + j * TX;
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc[j][0][k] = 0.f; acc[j][1][k] = 0.f; mu[j][k] = 0.f; is[j][k] = 0.f; sc[j][k] = 0.f; sh[j][k] = 0.f; }
    if (col < CH) {
      load8f(a.mean + col * 8, mu[j]); load8f(a.invstd + col * 8, is[j]);
      if (a.mask_mode == 2) { load8f(a.scale + col * 8, sc[j]); load8f(a.shift + col * 8, sh[j]); }
    }
  }
  const long long p0 = (long long)blockIdx.x * a.pix_per_block;
  const long long p1 = min(a.npix, p0 + a.pix_per_block);
#pragma unroll 2
  for (long long p = p0 + ty; p < p1; p += TY) {
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
      const int col = tx + j * TX;
      if (col < CH) {
        float yv[8], g[8];
        unpack8(*reinterpret_cast<const u32x4*>(a.y + p * a.ldy + col * 8), yv);
        masked_grad(a, p, col * 8, yv, g, sc[j], sh[j]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          acc[j][0][k] += g[k];
          acc[j][1][k] += g[k] * (yv[k] - mu[j][k]) * is[j][k];
        }
      }
    }
  }
  block_column_reduce<2, NCOL>(acc, tx, ty, TX, TY, CH, a.C, a.part + (size_t)blockIdx.x * 2 * a.C);
}

// The same reduction with the mask mode as a template parameter (no pooled gather): the loads of an iteration -- y, dz and
// the mask source -- are issued together instead of one per run-time branch with a wait after each.  Same summation order.
template <int NCOL, int MODE>
__global__ void __launch_bounds__(256) k_bn_bwd_reduce_s(BwdArgs a, int TX) {
  const int CH = a.C >> 3, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  float acc[NCOL][2][8];
  float mu[NCOL][8], is[NCOL][8], sc[NCOL][8], sh[NCOL][8];
#pragma unroll
  for (int j = 0; j < NCOL; ++j) {
    const int col = tx + j * TX;
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc[j][0][k] = 0.f; acc[j][1][k] = 0.f; mu[j][k] = 0.f; is[j][k] = 0.f; sc[j][k] = 0.f; sh[j][k] = 0.f; }
    if (col < CH) {
      load8f(a.mean + col * 8, mu[j]); load8f(a.invstd + col * 8, is[j]);
      if (MODE == 2) { load8f(a.scale + col * 8, sc[j]); load8f(a.shift + col * 8, sh[j]); }
    }
  }
  const long long p0 = (long long)blockIdx.x * a.pix_per_block;
  const long long p1 = min(a.npix, p0 + a.pix_per_block);
#pragma unroll 2
  for (long long p = p0 + ty; p < p1; p += TY) {
    u32x4 yq[NCOL], gq[NCOL], zq[NCOL];
    unsigned mq[NCOL];
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
      const int col = tx + j * TX;
      if (col < CH) {
        yq[j] = *reinterpret_cast<const u32x4*>(a.y + p * a.ldy + col * 8);
        gq[j] = *reinterpret_cast<const u32x4*>(a.dz + p * a.lddz + col * 8);
        if (MODE == 1) zq[j] = *reinterpret_cast<const u32x4*>(a.z + p * a.ldz + col * 8);
        if (MODE == 3) mq[j] = reinterpret_cast<const unsigned char*>(a.z)[p * a.ldz + col];
      }
    }
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
      const int col = tx + j * TX;
      if (col < CH) {
        float yv[8], g[8];
        unpack8(yq[j], yv);
        unpack8(gq[j], g);
        if (MODE == 1) {
          float zv[8];
          unpack8(zq[j], zv);
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] = zv[k] > 0.f ? g[k] : 0.f;
        } else if (MODE == 3) {
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] = ((mq[j] >> k) & 1u) ? g[k] : 0.f;
        } else if (MODE == 2) {
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] = (yv[k] * sc[j][k] + sh[j][k]) > 0.f ? g[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          acc[j][0][k] += g[k];
          acc[j][1][k] += g[k] * (yv[k] - mu[j][k]) * is[j][k];
        }
      }
    }
  }
  block_column_reduce<2, NCOL>(acc, tx, ty, TX, TY, CH, a.C, a.part + (size_t)blockIdx.x * 2 * a.C);
}

// Dual form: the masked gradient is shared by two BatchNorms (bn3 and the downsample BatchNorm of a projection block).
// sum(g) is common; sum(g * xhat) differs -- three running sums per channel, two partial tables (the second table's
// first row is the same sum(g): the finalize kernel is used unchanged for both).
template <int NCOL, int MODE>
__global__ void __launch_bounds__(256) k_bn_bwd_reduce_dual(BwdArgs a, int TX) {
  const int CH = a.C >> 3, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  float acc[NCOL][2][8], acc2[NCOL][2][8];
  float mu[NCOL][8], is[NCOL][8], mu2[NCOL][8], is2[NCOL][8];
#pragma unroll
  for (int j = 0; j < NCOL; ++j) {
    const int col = tx + j * TX;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc[j][0][k] = 0.f; acc[j][1][k] = 0.f; acc2[j][0][k] = 0.f; acc2[j][1][k] = 0.f;
      mu[j][k] = 0.f; is[j][k] = 0.f; mu2[j][k] = 0.f; is2[j][k] = 0.f;
    }
    if (col < CH) {
      load8f(a.mean + col * 8, mu[j]); load8f(a.invstd + col * 8, is[j]);
      load8f(a.mean2 + col * 8, mu2[j]); load8f(a.invstd2 + col * 8, is2[j]);
    }
  }
  const long long p0 = (long long)blockIdx.x * a.pix_per_block;
  const long long p1 = min(a.npix, p0 + a.pix_per_block);
#pragma unroll 2
  for (long long p = p0 + ty; p < p1; p += TY) {
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
      const int col = tx + j * TX;
      if (col < CH) {
        // every load of the iteration first (mask mode at compile time: see k_bn_bwd_reduce_s)
        const u32x4 yq = *reinterpret_cast<const u32x4*>(a.y + p * a.ldy + col * 8);
        const u32x4 y2q = *reinterpret_cast<const u32x4*>(a.y2 + p * a.ldy2 + col * 8);
        const u32x4 gq = *reinterpret_cast<const u32x4*>(a.dz + p * a.lddz + col * 8);
        u32x4 zq;
        unsigned mq = 0;
        if (MODE == 1) zq = *reinterpret_cast<const u32x4*>(a.z + p * a.ldz + col * 8);
        if (MODE == 3) mq = reinterpret_cast<const unsigned char*>(a.z)[p * a.ldz + col];
        float yv[8], y2v[8], g[8];
        unpack8(yq, yv);
        unpack8(y2q, y2v);
        unpack8(gq, g);
        if (MODE == 1) {
          float zv[8];
          unpack8(zq, zv);
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] = zv[k] > 0.f ? g[k] : 0.f;
        } else if (MODE == 3) {
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] = ((mq >> k) & 1u) ? g[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          acc[j][0][k] += g[k];
          acc[j][1][k] += g[k] * (yv[k] - mu[j][k]) * is[j][k];
          acc2[j][1][k] += g[k] * (y2v[k] - mu2[j][k]) * is2[j][k];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NCOL; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc2[j][0][k] = acc[j][0][k];
  block_column_reduce<2, NCOL>(acc, tx, ty, TX, TY, CH, a.C, a.part + (size_t)blockIdx.x * 2 * a.C);
  __syncthreads();                                     // block_column_reduce's LDS scratch is reused
  block_column_reduce<2, NCOL>(acc2, tx, ty, TX, TY, CH, a.C, a.part2 + (size_t)blockIdx.x * 2 * a.C);
}

// dgamma = sum dyh*xhat, dbeta = sum dyh;  dy = k1*dyh - k2 - xhat*k3 with
// k1 = gamma*invstd, k2 = k1*dbeta/n, k3 = k1*dgamma/n
template <int CPB>
__global__ void __launch_bounds__(1024) k_bn_bwd_finalize(const float* __restrict__ part, int rows, int C, float count,
                                                         const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         float* __restrict__ k1, float* __restrict__ k2,
                                                         float* __restrict__ k3) {
  constexpr int RL = 1024 / CPB;                 // see k_bn_finalize
  __shared__ float red[2][RL][CPB];
  const int tx = threadIdx.x % CPB, ty = threadIdx.x / CPB;
  const int c = blockIdx.x * CPB + tx;
  float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
  if (c < C) {
    int r = ty;
    for (; r + RL < rows; r += 2 * RL) {
      s0 += part[(size_t)r * 2 * C + c];
      q0 += part[(size_t)r * 2 * C + C + c];
      s1 += part[(size_t)(r + RL) * 2 * C + c];
      q1 += part[(size_t)(r + RL) * 2 * C + C + c];
    }
    for (; r < rows; r += RL) {
      s0 += part[(size_t)r * 2 * C + c];
      q0 += part[(size_t)r * 2 * C + C + c];
    }
  }
  red[0][ty][tx] = s0 + s1;
  red[1][ty][tx] = q0 + q1;
  __syncthreads();
  if (ty != 0 || c >= C) return;
  float s = 0.f, sx = 0.f;
#pragma unroll 16
  for (int j = 0; j < RL; ++j) { s += red[0][j][tx]; sx += red[1][j][tx]; }
  if (dgamma) dgamma[c] = sx;
  if (dbeta) dbeta[c] = s;
  const float g = gamma ? gamma[c] : 1.f;
  const float a1 = g * invstd[c];
  k1[c] = a1;
  k2[c] = a1 * s / count;
  k3[c] = a1 * sx / count;
}

__device__ __forceinline__ void bwd_apply_coeffs(const BwdArgs& a, int c8, float* k1, float* ka, float* kb) {
  float mu[8], is[8], k2[8], k3[8];
  load8f(a.mean + c8, mu); load8f(a.invstd + c8, is);
  load8f(a.k1 + c8, k1); load8f(a.k2 + c8, k2); load8f(a.k3 + c8, k3);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float t = is[k] * k3[k];
    ka[k] = -t;
    kb[k] = mu[k] * t - k2[k];
  }
}

template <bool FIXED_C>                                  // see k_bn_apply: 10 (14) of 14 (18) memory instructions were these
__global__ void __launch_bounds__(256) k_bn_bwd_apply(BwdArgs a) {
  const int CH = a.C >> 3;
  const long long total = a.npix * CH;
  // dy = k1*g - k2 - (y - mean)*invstd*k3 = k1*g + ka*y + kb with ka = -invstd*k3, kb = mean*invstd*k3 - k2: two FMAs per
  // value and three per-channel vectors live across the loop instead of five
  float k1[8], ka[8], kb[8], sc[8], sh[8];
  if (FIXED_C) {
    const int c8 = (int)((blockIdx.x * 256u + threadIdx.x) % CH) * 8;
    bwd_apply_coeffs(a, c8, k1, ka, kb);
    if (a.mask_mode == 2) { load8f(a.scale + c8, sc); load8f(a.shift + c8, sh); }
  }
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x;
  const long long dpix = FIXED_C ? (long long)gridDim.x * 256 / CH : 0;
  long long pix_f = i0 / (unsigned)CH;
  const int cc_f = (int)(i0 % (unsigned)CH) * 8;
  for (long long i = i0; FIXED_C ? pix_f < a.npix : i < total; i += (long long)gridDim.x * 256, pix_f += dpix) {
    const long long p = FIXED_C ? pix_f : i / CH;
    const int c8 = FIXED_C ? cc_f : (int)(i - p * CH) * 8;
    float yv[8], g[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.y + p * a.ldy + c8), yv);
    if (FIXED_C) masked_grad(a, p, c8, yv, g, sc, sh);
    else masked_grad(a, p, c8, yv, g);
    if (a.dres) *reinterpret_cast<u32x4*>(a.dres + p * a.lddres + c8) = pack8(g);
    if (!FIXED_C) bwd_apply_coeffs(a, c8, k1, ka, kb);
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = k1[k] * g[k] + (ka[k] * yv[k] + kb[k]);
    if (a.accumulate) {
      float old[8];
      unpack8(*reinterpret_cast<const u32x4*>(a.dy + p * a.lddy + c8), old);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += old[k];
    }
    *reinterpret_cast<u32x4*>(a.dy + p * a.lddy + c8) = pack8(o);
  }
}

// Dual apply: dy = k1*g + ka*y + kb and dy2 = k1b*g + ka2*y2 + kb2 from one read of the gradient and the mask.
__global__ void __launch_bounds__(256) k_bn_bwd_apply_dual(BwdArgs a) {
  const int CH = a.C >> 3;
  // the host rounds the grid so that the grid stride is a multiple of CH: a thread keeps its channel chunk
  const int c8 = (int)((blockIdx.x * 256u + threadIdx.x) % CH) * 8;
  float k1[8], ka[8], kb[8], k1b[8], ka2[8], kb2[8];
  bwd_apply_coeffs(a, c8, k1, ka, kb);
  {
    float mu[8], is[8], k2[8], k3[8];
    load8f(a.mean2 + c8, mu); load8f(a.invstd2 + c8, is);
    load8f(a.k1b + c8, k1b); load8f(a.k2b + c8, k2); load8f(a.k3b + c8, k3);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float t = is[k] * k3[k];
      ka2[k] = -t;
      kb2[k] = mu[k] * t - k2[k];
    }
  }
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x;
  const long long dpix = (long long)gridDim.x * 256 / CH;
  for (long long p = i0 / (unsigned)CH; p < a.npix; p += dpix) {
    float yv[8], y2v[8], g[8], o[8], o2[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.y + p * a.ldy + c8), yv);
    unpack8(*reinterpret_cast<const u32x4*>(a.y2 + p * a.ldy2 + c8), y2v);
    masked_grad(a, p, c8, yv, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o[k] = k1[k] * g[k] + (ka[k] * yv[k] + kb[k]);
      o2[k] = k1b[k] * g[k] + (ka2[k] * y2v[k] + kb2[k]);
    }
    *reinterpret_cast<u32x4*>(a.dy + p * a.lddy + c8) = pack8(o);
    *reinterpret_cast<u32x4*>(a.dy2 + p * a.lddy2 + c8) = pack8(o2);
  }
}

// ---- finalize fused into the consuming launch (round 3) ----------------------------------------------------------
// A training-mode BatchNorm used to be three launches in each direction: producer of the partial rows (the convolution's
// epilogue / the backward reduction) -> finalize (a 5-7 us kernel on C/16 workgroups) -> apply.  Inside the captured
// step every dependent launch on the main queue costs its own duration plus ~4.5 us of dispatch gap, and the two
// finalize kernels of the 53 BatchNorms were 127 of the step's 595 launches.  Here the finalize runs
// INSIDE the apply launch: the first `producers` workgroups of the grid finalize a slice of channels each (the same
// arithmetic, in a fixed order: bitwise reproducible), publish the per-channel coefficients and leave; every other
// workgroup is a consumer: it waits until all producers have published, then streams the tensor exactly as
// k_bn_apply / k_bn_bwd_apply do.
//   * No deadlock: producers never wait, and the dispatcher hands out the workgroups of a launch in index order on every
//     XCD, so a consumer can only be resident (and spinning) when the producers that precede it have been dispatched.
//     The spin still has an exit condition of its own: after ~1 s it sets the fault word (checked by the host: tests,
//     bench) and goes on.
//   * Coherence: the per-XCD L2s are not coherent with each other inside a launch.  Producers publish with agent-scope
//     (sc1, write-through) stores followed by s_waitcnt vmcnt(0) and ONE agent-scope atomic add per workgroup; consumers
//     poll with agent-scope loads and fetch the coefficients with agent-scope loads as well -- ONCE per workgroup, into
//     LDS (1024 workgroups x 256 lanes each fetching their own 64-128 bytes would hammer the 16-64 memory lines that hold
//     them).  No buffer_wbl2 / buffer_inv: the weight-gradient kernels of the side stream live off their L2 hits.
//   * The counter pair resets itself: the last consumer through the gate zeroes it (the next launch on that slot is
//     thousands of launches away), so a captured graph replays without a memset node.
// MEASURED RESULT (tools/bench_bn_fused.py, MI355X, captured graph): slower than the launch pairs on every ResNet-50
// shape -- 512 ch @7x7: 24.8 vs 6.7 us forward, 2048 ch @14x14: 76 vs 25 us; the step 24.5 vs 21.65 ms.  The premise was
// wrong: a dependent launch inside a hipGraph costs 1-2 us and the finalize kernel ~3 us (round 2's 5-7 us + 4.5 us gap
// were rocprofv3's own overhead), while store(sc1) -> atomic -> poll(sc1) -> atomic -> load(sc1) is five trips to the
// memory side of an eight-XCD chip, 13-18 us in all.  The host layer therefore keeps the launch pairs
// (ops.BN_FUSED = False); the fused entry points stay as the tested, bit-identical record of the experiment.
struct FuseSync {
  unsigned* ctr;       // [0] producers that have published, [1] consumers that have passed the gate; both zero between launches
  unsigned* fault;     // set to 1 by a consumer whose wait timed out
  int producers;       // workgroups [0, producers) finalize, [producers, gridDim.x) consume
};

__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void fuse_publish(const FuseSync& s) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores have reached memory
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(s.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void fuse_wait(const FuseSync& s) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(s.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s.producers) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 20)) { atomicOr(s.fault, 1u); break; }        // exit condition: ~1 s
    }
    const unsigned consumers = gridDim.x - (unsigned)s.producers;
    const unsigned d = __hip_atomic_fetch_add(s.ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == consumers - 1) {                             // everyone is through: leave the pair zeroed for the next launch
      __hip_atomic_store(s.ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(s.ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
}

// n floats (n % 4 == 0, src 16-byte aligned) published by producers of THIS launch -> LDS, agent scope.  All of a lane's
// loads are in flight together and waited for once: as a loop of atomic dword loads (which the compiler keeps in order,
// one memory round trip each) the staging alone cost up to 30 us per launch at C = 2048 -- the first version of this
// fusion made the step 13 % SLOWER.  Loads and their wait are one asm statement (the outputs are not valid before it).
__device__ __forceinline__ void stage_agent(float* dst, const float* src, int n) {
  const int nq = n >> 2;
  for (int base = 0; base < nq; base += 4 * 256) {
    float4 v0, v1, v2, v3;
    const int q0 = base + threadIdx.x, q1 = q0 + 256, q2 = q0 + 512, q3 = q0 + 768;
    const float* p0 = src + 4 * (q0 < nq ? q0 : 0);
    const float* p1 = src + 4 * (q1 < nq ? q1 : 0);
    const float* p2 = src + 4 * (q2 < nq ? q2 : 0);
    const float* p3 = src + 4 * (q3 < nq ? q3 : 0);
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
    if (q0 < nq) reinterpret_cast<float4*>(dst)[q0] = v0;
    if (q1 < nq) reinterpret_cast<float4*>(dst)[q1] = v1;
    if (q2 < nq) reinterpret_cast<float4*>(dst)[q2] = v2;
    if (q3 < nq) reinterpret_cast<float4*>(dst)[q3] = v3;
  }
}

// Forward finalize of FCPB channels per 256-thread workgroup (64 row lanes each): the arithmetic of k_bn_finalize.
constexpr int FCPB = 4, FRL = 256 / FCPB;
struct FinFwd {
  const float* part; int rows, C, ldp; float count;
  const float* gamma; const float* beta; float eps, momentum;
  float* rmean; float* rvar; float* mean; float* invstd; float* scale; float* shift;
  int nblk;                                               // ceil(C / FCPB); 0: nothing to finalize
  // merged form (DenseNet, see k_bn_finalize): channels [seg_c0, seg_c0 + seg_c) are still the partial rows
  // seg_part [seg_rows][2][seg_c] of the convolution that produced them; summed here and written into table_out
  const float* seg_part = nullptr; int seg_rows = 0, seg_c0 = 0, seg_c = 0; float* table_out = nullptr;
};

__device__ __forceinline__ void fin_fwd_body(const FinFwd& f, int blk, float* red /* [2][FRL][FCPB] */) {
  const int tx = threadIdx.x % FCPB, ty = threadIdx.x / FCPB;
  const int c = blk * FCPB + tx;
  float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
  const bool seg = f.seg_part != nullptr && c >= f.seg_c0 && c < f.seg_c0 + f.seg_c;
  if (c < f.C) {
    const float* src = seg ? f.seg_part + (c - f.seg_c0) : f.part + c;
    const size_t rs = seg ? (size_t)2 * f.seg_c : (size_t)2 * f.ldp, qoff = seg ? f.seg_c : f.ldp;
    const int nrows = seg ? f.seg_rows : f.rows;
    int r = ty;
#pragma unroll 2
    for (; r + FRL < nrows; r += 2 * FRL) {
      s0 += src[(size_t)r * rs];
      q0 += src[(size_t)r * rs + qoff];
      s1 += src[(size_t)(r + FRL) * rs];
      q1 += src[(size_t)(r + FRL) * rs + qoff];
    }
    for (; r < nrows; r += FRL) {
      s0 += src[(size_t)r * rs];
      q0 += src[(size_t)r * rs + qoff];
    }
  }
  red[(0 * FRL + ty) * FCPB + tx] = s0 + s1;
  red[(1 * FRL + ty) * FCPB + tx] = q0 + q1;
  __syncthreads();
  if (ty == 0 && c < f.C) {
    float s = 0.f, ss = 0.f;
#pragma unroll 16
    for (int j = 0; j < FRL; ++j) { s += red[(0 * FRL + j) * FCPB + tx]; ss += red[(1 * FRL + j) * FCPB + tx]; }
    if (seg) { f.table_out[c] = s; f.table_out[f.ldp + c] = ss; }
    const float mean = s / f.count;
    float var = ss / f.count - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float invstd = rsqrtf(var + f.eps);
    const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    f.mean[c] = mean;                                     // read by later launches only
    f.invstd[c] = invstd;
    st_agent(f.scale + c, g * invstd);                    // read by the consumers of this launch
    st_agent(f.shift + c, b - mean * g * invstd);
    if (f.rmean) {
      f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * mean;
      const float unb = f.count > 1.f ? var * f.count / (f.count - 1.f) : var;
      f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * unb;
    }
  }
}

// k_bn_apply with up to two BatchNorms finalized by its first workgroups: f1 = the BatchNorm applied to y, f2 = the
// BatchNorm of the residual (projection shortcut: a.rscale / a.rshift), nblk 0 when absent or already final.
template <bool FIXED_C>
__global__ void __launch_bounds__(256) k_bn_apply_fused(ApplyArgs a, FinFwd f1, FinFwd f2, FuseSync s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];    // producers: 2*FRL*FCPB floats; consumers: up to 4 C
  if ((int)blockIdx.x < s.producers) {
    const int b = blockIdx.x;
    if (b < f1.nblk) fin_fwd_body(f1, b, lds); else fin_fwd_body(f2, b - f1.nblk, lds);
    fuse_publish(s);
    return;
  }
  const unsigned bid = blockIdx.x - s.producers, nblk = gridDim.x - s.producers;
  const int C = a.C, CH = C >> 3;
  const bool res_bn = a.res && a.rscale;
  fuse_wait(s);
  float* l_sc = lds; float* l_sh = lds + C; float* l_rs = lds + 2 * C; float* l_rh = lds + 3 * C;
  // BNState keeps scale and shift in consecutive rows: one staging pass (one memory round trip) for both
  if (f1.nblk) {
    if (a.shift == a.scale + C) stage_agent(l_sc, a.scale, 2 * C);
    else { stage_agent(l_sc, a.scale, C); stage_agent(l_sh, a.shift, C); }
  } else { for (int i = threadIdx.x; i < C; i += 256) { l_sc[i] = a.scale[i]; l_sh[i] = a.shift[i]; } }
  if (res_bn) {
    if (f2.nblk) {
      if (a.rshift == a.rscale + C) stage_agent(l_rs, a.rscale, 2 * C);
      else { stage_agent(l_rs, a.rscale, C); stage_agent(l_rh, a.rshift, C); }
    } else { for (int i = threadIdx.x; i < C; i += 256) { l_rs[i] = a.rscale[i]; l_rh[i] = a.rshift[i]; } }
  }
  __syncthreads();
  const long long total = a.npix * CH;
  float sc[8], sh[8], rs[8], rh[8];
  if (FIXED_C) {
    const int cc = (int)((bid * 256u + threadIdx.x) % CH) * 8;
    load8f(l_sc + cc, sc);
    load8f(l_sh + cc, sh);
    if (res_bn) { load8f(l_rs + cc, rs); load8f(l_rh + cc, rh); }
  }
  const unsigned i0 = bid * 256u + threadIdx.x;
  const long long dpix = FIXED_C ? (long long)nblk * 256 / CH : 0;
  long long pix_f = i0 / (unsigned)CH;
  const int cc_f = (int)(i0 % (unsigned)CH) * 8;
  for (long long i = i0; FIXED_C ? pix_f < a.npix : i < total; i += (long long)nblk * 256, pix_f += dpix) {
    const long long pix = FIXED_C ? pix_f : i / CH;
    const int cc = FIXED_C ? cc_f : (int)(i - pix * CH) * 8;
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.y + pix * a.ldy + cc), f);
    if (!FIXED_C) { load8f(l_sc + cc, sc); load8f(l_sh + cc, sh); }
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = f[k] * sc[k] + sh[k];
    if (a.res) {
      float r[8];
      unpack8(*reinterpret_cast<const u32x4*>(a.res + pix * a.ldr + cc), r);
      if (a.rscale) {
        if (!FIXED_C) { load8f(l_rs + cc, rs); load8f(l_rh + cc, rh); }
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = r[k] * rs[k] + rh[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] += r[k];
    }
    if (a.relu) {
      if (a.relu_mask) {
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) m |= (f[k] > 0.f ? 1u : 0u) << k;
        a.relu_mask[pix * CH + (cc >> 3)] = (unsigned char)m;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k], 0.f);
    }
    const u32x4 zb = pack8(f);
    *reinterpret_cast<u32x4*>(a.z + pix * a.ldz + cc) = zb;
    if (a.z8) {
      float r[8];
      unpack8(zb, r);
      uint2 o;
      o.x = pack_e4m3x4(r[0], r[1], r[2], r[3]);
      o.y = pack_e4m3x4(r[4], r[5], r[6], r[7]);
      *reinterpret_cast<uint2*>(a.z8 + pix * a.ldz8 + cc) = o;
    }
  }
}

// Backward finalize (the arithmetic of k_bn_bwd_finalize) on FCPB channels per 256-thread workgroup.
struct FinBwd {
  const float* part; int rows, C; float count;
  const float* gamma; const float* invstd;
  float* dgamma; float* dbeta; float* k1; float* k2; float* k3;
  int nblk;
};

__device__ __forceinline__ void fin_bwd_body(const FinBwd& f, int blk, float* red) {
  const int tx = threadIdx.x % FCPB, ty = threadIdx.x / FCPB;
  const int c = blk * FCPB + tx;
  float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
  if (c < f.C) {
    const size_t rs = (size_t)2 * f.C;
    int r = ty;
#pragma unroll 2
    for (; r + FRL < f.rows; r += 2 * FRL) {
      s0 += f.part[(size_t)r * rs + c];
      q0 += f.part[(size_t)r * rs + f.C + c];
      s1 += f.part[(size_t)(r + FRL) * rs + c];
      q1 += f.part[(size_t)(r + FRL) * rs + f.C + c];
    }
    for (; r < f.rows; r += FRL) {
      s0 += f.part[(size_t)r * rs + c];
      q0 += f.part[(size_t)r * rs + f.C + c];
    }
  }
  red[(0 * FRL + ty) * FCPB + tx] = s0 + s1;
  red[(1 * FRL + ty) * FCPB + tx] = q0 + q1;
  __syncthreads();
  if (ty == 0 && c < f.C) {
    float s = 0.f, sx = 0.f;
#pragma unroll 16
    for (int j = 0; j < FRL; ++j) { s += red[(0 * FRL + j) * FCPB + tx]; sx += red[(1 * FRL + j) * FCPB + tx]; }
    if (f.dgamma) f.dgamma[c] = sx;
    if (f.dbeta) f.dbeta[c] = s;
    const float g = f.gamma ? f.gamma[c] : 1.f;
    const float a1 = g * f.invstd[c];
    st_agent(f.k1 + c, a1);
    st_agent(f.k2 + c, a1 * s / f.count);
    st_agent(f.k3 + c, a1 * sx / f.count);
  }
}

// coefficient vectors of the backward apply from LDS copies of k1/k2/k3 (mean / invstd come from earlier launches)
__device__ __forceinline__ void bwd_coeffs_lds(const float* mean, const float* invstd, const float* l_k1, const float* l_k2,
                                               const float* l_k3, int c8, float* k1, float* ka, float* kb) {
  float mu[8], is[8], k2[8], k3[8];
  load8f(mean + c8, mu); load8f(invstd + c8, is);
  load8f(l_k1 + c8, k1); load8f(l_k2 + c8, k2); load8f(l_k3 + c8, k3);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float t = is[k] * k3[k];
    ka[k] = -t;
    kb[k] = mu[k] * t - k2[k];
  }
}

// k_bn_bwd_apply (no pooled gather) with its finalize in the first workgroups.
template <bool FIXED_C>
__global__ void __launch_bounds__(256) k_bn_bwd_apply_fused(BwdArgs a, FinBwd f1, FuseSync s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x < s.producers) {
    fin_bwd_body(f1, blockIdx.x, lds);
    fuse_publish(s);
    return;
  }
  const unsigned bid = blockIdx.x - s.producers, nblk = gridDim.x - s.producers;
  const int C = a.C, CH = C >> 3;
  fuse_wait(s);
  float* l_k1 = lds; float* l_k2 = lds + C; float* l_k3 = lds + 2 * C;
  if (a.k2 == a.k1 + C && a.k3 == a.k1 + 2 * C) stage_agent(l_k1, a.k1, 3 * C);     // consecutive rows of one scratch buffer
  else { stage_agent(l_k1, a.k1, C); stage_agent(l_k2, a.k2, C); stage_agent(l_k3, a.k3, C); }
  __syncthreads();
  const long long total = a.npix * CH;
  float k1[8], ka[8], kb[8], sc[8], sh[8];
  if (FIXED_C) {
    const int c8 = (int)((bid * 256u + threadIdx.x) % CH) * 8;
    bwd_coeffs_lds(a.mean, a.invstd, l_k1, l_k2, l_k3, c8, k1, ka, kb);
    if (a.mask_mode == 2) { load8f(a.scale + c8, sc); load8f(a.shift + c8, sh); }
  }
  const unsigned i0 = bid * 256u + threadIdx.x;
  const long long dpix = FIXED_C ? (long long)nblk * 256 / CH : 0;
  long long pix_f = i0 / (unsigned)CH;
  const int cc_f = (int)(i0 % (unsigned)CH) * 8;
  for (long long i = i0; FIXED_C ? pix_f < a.npix : i < total; i += (long long)nblk * 256, pix_f += dpix) {
    const long long p = FIXED_C ? pix_f : i / CH;
    const int c8 = FIXED_C ? cc_f : (int)(i - p * CH) * 8;
    float yv[8], g[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.y + p * a.ldy + c8), yv);
    if (FIXED_C) masked_grad(a, p, c8, yv, g, sc, sh);
    else masked_grad(a, p, c8, yv, g);
    if (a.dres) *reinterpret_cast<u32x4*>(a.dres + p * a.lddres + c8) = pack8(g);
    if (!FIXED_C) bwd_coeffs_lds(a.mean, a.invstd, l_k1, l_k2, l_k3, c8, k1, ka, kb);
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = k1[k] * g[k] + (ka[k] * yv[k] + kb[k]);
    if (a.accumulate) {
      float old[8];
      unpack8(*reinterpret_cast<const u32x4*>(a.dy + p * a.lddy + c8), old);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += old[k];
    }
    *reinterpret_cast<u32x4*>(a.dy + p * a.lddy + c8) = pack8(o);
  }
}

// Dual form (projection blocks): both finalizes in the first workgroups, then k_bn_bwd_apply_dual's loop.
__global__ void __launch_bounds__(256) k_bn_bwd_apply_dual_fused(BwdArgs a, FinBwd f1, FinBwd f2, FuseSync s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x < s.producers) {
    const int b = blockIdx.x;
    if (b < f1.nblk) fin_bwd_body(f1, b, lds); else fin_bwd_body(f2, b - f1.nblk, lds);
    fuse_publish(s);
    return;
  }
  const unsigned bid = blockIdx.x - s.producers, nblk = gridDim.x - s.producers;
  const int C = a.C, CH = C >> 3;
  fuse_wait(s);
  stage_agent(lds, a.k1, 6 * C);                       // k6: [k1, k2, k3, k1b, k2b, k3b] rows of one scratch buffer
  __syncthreads();
  const int c8 = (int)((bid * 256u + threadIdx.x) % CH) * 8;
  float k1[8], ka[8], kb[8], k1b[8], ka2[8], kb2[8];
  bwd_coeffs_lds(a.mean, a.invstd, lds, lds + C, lds + 2 * C, c8, k1, ka, kb);
  bwd_coeffs_lds(a.mean2, a.invstd2, lds + 3 * C, lds + 4 * C, lds + 5 * C, c8, k1b, ka2, kb2);
  const unsigned i0 = bid * 256u + threadIdx.x;
  const long long dpix = (long long)nblk * 256 / CH;
  for (long long p = i0 / (unsigned)CH; p < a.npix; p += dpix) {
    float yv[8], y2v[8], g[8], o[8], o2[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.y + p * a.ldy + c8), yv);
    unpack8(*reinterpret_cast<const u32x4*>(a.y2 + p * a.ldy2 + c8), y2v);
    masked_grad(a, p, c8, yv, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o[k] = k1[k] * g[k] + (ka[k] * yv[k] + kb[k]);
      o2[k] = k1b[k] * g[k] + (ka2[k] * y2v[k] + kb2[k]);
    }
    *reinterpret_cast<u32x4*>(a.dy + p * a.lddy + c8) = pack8(o);
    *reinterpret_cast<u32x4*>(a.dy2 + p * a.lddy2 + c8) = pack8(o2);
  }
}

// ---- pooled form on aligned 2x2 patches (even H and W: the stems) ------------------------------------------------
// The per-pixel gather above re-reads every pool window for each of the up-to-nine pixels under it and pays two integer
// divisions per pixel; it is latency-bound (the stem's pair ran no faster than max-pool backward + plain BN backward).
// An aligned 2x2 patch of the pool input (rows 2a,2a+1 / columns 2b,2b+1) lies under exactly the four windows
// (a,b) (a,b+1) (a+1,b) (a+1,b+1): one thread loads those four (index, gradient) pairs once and produces the four
// pixel gradients, in the same summation order as k_maxpool_bwd_idx, rounded to bf16 like the tensor it replaces.
__device__ __forceinline__ void pooled_patch_grad(const BwdArgs& a, unsigned n, unsigned pa, unsigned pb, int c8, int OH, int OW,
                                                  float (&g)[4][8]) {
  unsigned code[4][8];
  float d[4][8];
#pragma unroll
  for (int wi = 0; wi < 4; ++wi) {
    const unsigned oh = pa + (wi >> 1), ow = pb + (wi & 1);
    const bool ok = (int)oh < OH && (int)ow < OW;
    const size_t o = ok ? (size_t)(n * OH + oh) * OW + ow : 0;
    const uint2 am2 = *reinterpret_cast<const uint2*>(a.pool_idx + o * a.C + c8);
    unpack8(*reinterpret_cast<const u32x4*>(a.dz + o * a.lddz + c8), d[wi]);
#pragma unroll
    for (int k = 0; k < 8; ++k) code[wi][k] = ok ? ((k < 4 ? am2.x : am2.y) >> (8 * (k & 3))) & 0xffu : 0xffu;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // window position r*3+s of pixel (h,w) inside window (oh,ow): r = h - (2oh-1), s = w - (2ow-1)
    g[0][k] = 0.f + (code[0][k] == 4u ? d[0][k] : 0.f);
    g[1][k] = (0.f + (code[0][k] == 5u ? d[0][k] : 0.f)) + (code[1][k] == 3u ? d[1][k] : 0.f);
    g[2][k] = (0.f + (code[0][k] == 7u ? d[0][k] : 0.f)) + (code[2][k] == 1u ? d[2][k] : 0.f);
    g[3][k] = (((0.f + (code[0][k] == 8u ? d[0][k] : 0.f)) + (code[1][k] == 6u ? d[1][k] : 0.f)) +
               (code[2][k] == 2u ? d[2][k] : 0.f)) + (code[3][k] == 0u ? d[3][k] : 0.f);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) unpack8(pack8(g[q]), g[q]);
}

// reduce over patches: a.npix = number of patches, a.pix_per_block patches per block; C <= 8*TX (one column per thread)
__global__ void __launch_bounds__(256, 3) k_bn_bwd_reduce_pool2(BwdArgs a, int TX) {
  const int CH = a.C >> 3, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int OH = a.pH >> 1, OW = a.pW >> 1;
  float acc[1][2][8];
  float mu[8], is[8], sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { acc[0][0][k] = 0.f; acc[0][1][k] = 0.f; mu[k] = 0.f; is[k] = 0.f; sc[k] = 0.f; sh[k] = 0.f; }
  if (tx < CH) {
    load8f(a.mean + tx * 8, mu); load8f(a.invstd + tx * 8, is);
    if (a.mask_mode == 2) { load8f(a.scale + tx * 8, sc); load8f(a.shift + tx * 8, sh); }
  }
  const long long p0 = (long long)blockIdx.x * a.pix_per_block;
  const long long p1 = min(a.npix, p0 + a.pix_per_block);
  for (long long pp = p0 + ty; pp < p1; pp += TY) {
    if (tx < CH) {
      const unsigned up = (unsigned)pp;
      const unsigned pb = up % OW, t = up / OW, pa = t % OH, n = t / OH;
      float g[4][8];
      pooled_patch_grad(a, n, pa, pb, tx * 8, OH, OW, g);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t p = ((size_t)n * a.pH + 2 * pa + (q >> 1)) * a.pW + 2 * pb + (q & 1);
        float yv[8];
        unpack8(*reinterpret_cast<const u32x4*>(a.y + p * a.ldy + tx * 8), yv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float gk = (a.mask_mode == 2 && !(yv[k] * sc[k] + sh[k] > 0.f)) ? 0.f : g[q][k];
          acc[0][0][k] += gk;
          acc[0][1][k] += gk * (yv[k] - mu[k]) * is[k];
        }
      }
    }
  }
  block_column_reduce<2, 1>(acc, tx, ty, TX, TY, CH, a.C, a.part + (size_t)blockIdx.x * 2 * a.C);
}

__global__ void __launch_bounds__(256, 4) k_bn_bwd_apply_pool2(BwdArgs a) {
  const int CH = a.C >> 3, OH = a.pH >> 1, OW = a.pW >> 1;
  const long long total = a.npix * CH;                    // a.npix = number of patches
  // the grid stride is a multiple of CH when CH divides 256 (the stems: CH = 8): a thread then keeps its channel chunk
  // for the whole loop and the seven per-channel vectors are loaded once
  const bool fixed_c = (256 % CH) == 0;
  float k1[8], ka[8], kb[8], sc[8], sh[8];
  if (fixed_c) {
    const int c8 = (int)(threadIdx.x % CH) * 8;
    bwd_apply_coeffs(a, c8, k1, ka, kb);
    if (a.mask_mode == 2) { load8f(a.scale + c8, sc); load8f(a.shift + c8, sh); }
  }
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const unsigned ui = (unsigned)i;
    const int c8 = (int)(ui % CH) * 8;
    unsigned t = ui / CH;
    const unsigned pb = t % OW; t /= OW;
    const unsigned pa = t % OH, n = t / OH;
    float g[4][8];
    pooled_patch_grad(a, n, pa, pb, c8, OH, OW, g);
    if (!fixed_c) {
      bwd_apply_coeffs(a, c8, k1, ka, kb);
      if (a.mask_mode == 2) { load8f(a.scale + c8, sc); load8f(a.shift + c8, sh); }
    }
    // the four BatchNorm inputs of the patch up front: a load issued after the previous pixel's store waits for it (the
    // compiler cannot prove y and dy apart) -- four memory round trips per patch instead of one
    u32x4 yq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t p = ((size_t)n * a.pH + 2 * pa + (q >> 1)) * a.pW + 2 * pb + (q & 1);
      yq[q] = *reinterpret_cast<const u32x4*>(a.y + p * a.ldy + c8);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t p = ((size_t)n * a.pH + 2 * pa + (q >> 1)) * a.pW + 2 * pb + (q & 1);
      float yv[8], o[8];
      unpack8(yq[q], yv);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float gk = (a.mask_mode == 2 && !(yv[k] * sc[k] + sh[k] > 0.f)) ? 0.f : g[q][k];
        o[k] = k1[k] * gk + (ka[k] * yv[k] + kb[k]);
      }
      *reinterpret_cast<u32x4*>(a.dy + p * a.lddy + c8) = pack8(o);
    }
  }
}

// x [N,H,W,*] -> y [N,H/2,W/2,*]: the pixels (2h, 2w) a stride-2 1x1 convolution reads, made dense ("bn3 as algebra" for a
// strided projection shortcut needs x_s as a GEMM operand)
__global__ void __launch_bounds__(256) k_subsample2(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int N,
                                                    int H, int W, int C) {
  const int OH = H >> 1, OW = W >> 1, CH = C >> 3;
  const long long total = (long long)N * OH * OW * CH;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % CH);
    long long t = i / CH;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    const size_t src = ((size_t)(n * H + 2 * oh) * W + 2 * ow) * ldx + cc * 8;
    const size_t dst = ((size_t)(n * OH + oh) * OW + ow) * ldy + cc * 8;
    *reinterpret_cast<u32x4*>(y + dst) = *reinterpret_cast<const u32x4*>(x + src);
  }
}

// ------------------------------------------------------------------ max pool 3x3 / stride 2 / pad 1
__global__ void __launch_bounds__(256) k_maxpool_fwd(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                     unsigned char* __restrict__ idx, int N, int H, int W, int C) {
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1, CH = C >> 3;
  const long long total = (long long)N * OH * OW * CH;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % CH) * 8;
    long long t = i / CH;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    float m[8];
    unsigned am[8];                      // window position (r*3+s) of the FIRST maximum, the element torch selects
#pragma unroll
    for (int k = 0; k < 8; ++k) { m[k] = -INFINITY; am[k] = 0; }
    for (int r = 0; r < 3; ++r) {
      const int h = oh * 2 - 1 + r;
      if (h < 0 || h >= H) continue;
      for (int s = 0; s < 3; ++s) {
        const int w = ow * 2 - 1 + s;
        if (w < 0 || w >= W) continue;
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + ((size_t)(n * H + h) * W + w) * ldx + cc), f);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (f[k] > m[k]) { m[k] = f[k]; am[k] = r * 3 + s; }
      }
    }
    *reinterpret_cast<u32x4*>(y + ((size_t)(n * OH + oh) * OW + ow) * ldy + cc) = pack8(m);
    if (idx) {
      uint2 o;
      o.x = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
      o.y = am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24);
      *reinterpret_cast<uint2*>(idx + ((size_t)(n * OH + oh) * OW + ow) * C + cc) = o;
    }
  }
}

// gather form: every input pixel collects the gradient of each window whose FIRST maximum (row-major
// scan, the element torch's max_pool2d backward selects) it is.
__global__ void __launch_bounds__(256) k_maxpool_bwd(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ dy,
                                                     int lddy, bf16_t* __restrict__ dx, int lddx, int N, int H, int W, int C) {
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1, CH = C >> 3;
  const long long total = (long long)N * H * W * CH;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % CH) * 8;
    long long t = i / CH;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float me[8], g[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + ((size_t)(n * H + h) * W + w) * ldx + cc), me);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = 0.f;
    // windows (oh,ow) containing (h,w): oh*2-1 <= h <= oh*2+1
    const int oh_lo = max(0, (h) / 2), oh_hi = min(OH - 1, (h + 1) / 2);
    const int ow_lo = max(0, (w) / 2), ow_hi = min(OW - 1, (w + 1) / 2);
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        // is (h,w) the first max of window (oh,ow)?
        bool first[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) first[k] = true;
        for (int r = 0; r < 3; ++r) {
          const int hh = oh * 2 - 1 + r;
          if (hh < 0 || hh >= H) continue;
          for (int s = 0; s < 3; ++s) {
            const int ww = ow * 2 - 1 + s;
            if (ww < 0 || ww >= W || (hh == h && ww == w)) continue;
            float f[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + ((size_t)(n * H + hh) * W + ww) * ldx + cc), f);
            const bool before = (hh < h) || (hh == h && ww < w);
#pragma unroll
            for (int k = 0; k < 8; ++k) first[k] = first[k] && (before ? (f[k] < me[k]) : (f[k] <= me[k]));
          }
        }
        float d[8];
        unpack8(*reinterpret_cast<const u32x4*>(dy + ((size_t)(n * OH + oh) * OW + ow) * lddy + cc), d);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] += first[k] ? d[k] : 0.f;
      }
    *reinterpret_cast<u32x4*>(dx + ((size_t)(n * H + h) * W + w) * lddx + cc) = pack8(g);
  }
}

// BatchNorm-apply (+ReLU) + max pool in one pass (the stems): pools z = bf16(relu(y*scale + shift)) -- the tensor
// k_bn_apply would have stored -- without storing it: the training backward needs y and the argmax codes only.
__global__ void __launch_bounds__(256) k_bn_act_maxpool_fwd(const bf16_t* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu, bf16_t* __restrict__ out,
                                                            int ldo, unsigned char* __restrict__ idx, int N, int H, int W, int C) {
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1, CH = C >> 3;
  const long long total = (long long)N * OH * OW * CH;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const unsigned ui = (unsigned)i;
    const int cc = (int)(ui % CH) * 8;
    unsigned t = ui / CH;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    float m[8], sc[8], sh[8];
    unsigned am[8];
    load8f(scale + cc, sc);
    load8f(shift + cc, sh);
#pragma unroll
    for (int k = 0; k < 8; ++k) { m[k] = -INFINITY; am[k] = 0; }
    for (int r = 0; r < 3; ++r) {
      const int h = oh * 2 - 1 + r;
      if (h < 0 || h >= H) continue;
      for (int s_ = 0; s_ < 3; ++s_) {
        const int w = ow * 2 - 1 + s_;
        if (w < 0 || w >= W) continue;
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(y + ((size_t)(n * H + h) * W + w) * ldy + cc), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = f[k] * sc[k] + sh[k];
        if (relu) {
#pragma unroll
          for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k], 0.f);
        }
        unpack8(pack8(f), f);                          // the stored precision of z
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (f[k] > m[k]) { m[k] = f[k]; am[k] = r * 3 + s_; }
      }
    }
    *reinterpret_cast<u32x4*>(out + ((size_t)(n * OH + oh) * OW + ow) * ldo + cc) = pack8(m);
    if (idx) {
      uint2 o;
      o.x = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
      o.y = am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24);
      *reinterpret_cast<uint2*>(idx + ((size_t)(n * OH + oh) * OW + ow) * C + cc) = o;
    }
  }
}

// index form: the forward stored, per (window, channel), which of its 9 positions was the first maximum
__global__ void __launch_bounds__(256) k_maxpool_bwd_idx(const unsigned char* __restrict__ idx, const bf16_t* __restrict__ dy,
                                                         int lddy, bf16_t* __restrict__ dx, int lddx, int N, int H, int W, int C) {
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1, CH = C >> 3;
  const long long total = (long long)N * H * W * CH;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % CH) * 8;
    long long t = i / CH;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = 0.f;
    const int oh_lo = h / 2, oh_hi = min(OH - 1, (h + 1) / 2);
    const int ow_lo = w / 2, ow_hi = min(OW - 1, (w + 1) / 2);
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const unsigned code = (unsigned)((h - (oh * 2 - 1)) * 3 + (w - (ow * 2 - 1)));
        const size_t o = (size_t)(n * OH + oh) * OW + ow;
        const uint2 a = *reinterpret_cast<const uint2*>(idx + o * C + cc);
        float d[8];
        unpack8(*reinterpret_cast<const u32x4*>(dy + o * lddy + cc), d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned am = ((k < 4 ? a.x : a.y) >> (8 * (k & 3))) & 0xffu;
          g[k] += am == code ? d[k] : 0.f;
        }
      }
    *reinterpret_cast<u32x4*>(dx + ((size_t)(n * H + h) * W + w) * lddx + cc) = pack8(g);
  }
}

// ------------------------------------------------------------------ average pool 2x2 / stride 2
__global__ void __launch_bounds__(256) k_avgpool_fwd(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                     int N, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2, CH = C >> 3;
  const long long total = (long long)N * OH * OW * CH;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % CH) * 8;
    long long t = i / CH;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    for (int r = 0; r < 2; ++r)
      for (int q = 0; q < 2; ++q) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + ((size_t)(n * H + oh * 2 + r) * W + ow * 2 + q) * ldx + cc), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += f[k];
      }
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] *= 0.25f;
    *reinterpret_cast<u32x4*>(y + ((size_t)(n * OH + oh) * OW + ow) * ldy + cc) = pack8(s);
  }
}

__global__ void __launch_bounds__(256) k_avgpool_bwd(const bf16_t* __restrict__ dy, int lddy, bf16_t* __restrict__ dx, int lddx,
                                                     int N, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2, CH = C >> 3;
  const long long total = (long long)N * H * W * CH;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % CH) * 8;
    long long t = i / CH;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = 0.f;
    if (h / 2 < OH && w / 2 < OW) {
      unpack8(*reinterpret_cast<const u32x4*>(dy + ((size_t)(n * OH + h / 2) * OW + w / 2) * lddy + cc), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] *= 0.25f;
    }
    *reinterpret_cast<u32x4*>(dx + ((size_t)(n * H + h) * W + w) * lddx + cc) = pack8(f);
  }
}

// ------------------------------------------------------------------ head: bn_end + sigmoid
// y [npix][ldy] bf16 (first C channels valid) -> out [npix][C] fp32 = sigmoid(scale*y + shift)
__global__ void k_head_fwd(const bf16_t* __restrict__ y, int ldy, const float* __restrict__ scale,
                           const float* __restrict__ shift, float* __restrict__ out, long long npix, int C) {
  const long long total = npix * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / C;
    const int c = (int)(i - p * C);
    const float v = bf16_to_f32(y[p * ldy + c]) * scale[c] + shift[c];
    out[i] = 1.f / (1.f + __expf(-v));
  }
}

// one workgroup per channel: sigmoid backward + BN backward (reduce and apply fused; the head is tiny)
__global__ void __launch_bounds__(256) k_head_bwd(const float* __restrict__ dout, const float* __restrict__ out,
                                                  const bf16_t* __restrict__ y, int ldy, const float* __restrict__ gamma,
                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                  bf16_t* __restrict__ dy, int lddy, float* __restrict__ dgamma,
                                                  float* __restrict__ dbeta, long long npix, int C) {
  __shared__ float red[2][4];
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (c >= C) {   // padding channels of the 32-wide head buffer carry no gradient
    for (long long p = threadIdx.x; p < npix; p += 256) dy[p * lddy + c] = 0;
    return;
  }
  const float mu = mean[c], is = invstd[c];
  float s = 0.f, sx = 0.f;
  for (long long p = threadIdx.x; p < npix; p += 256) {
    const float o = out[p * C + c];
    const float g = dout[p * C + c] * o * (1.f - o);
    s += g;
    sx += g * (bf16_to_f32(y[p * ldy + c]) - mu) * is;
  }
  s = wave_sum(s); sx = wave_sum(sx);
  if (lane == 0) { red[0][wid] = s; red[1][wid] = sx; }
  __syncthreads();
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  sx = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  if (threadIdx.x == 0) { dgamma[c] = sx; dbeta[c] = s; }
  const float k1 = gamma[c] * is, n = (float)npix;
  for (long long p = threadIdx.x; p < npix; p += 256) {
    const float o = out[p * C + c];
    const float g = dout[p * C + c] * o * (1.f - o);
    const float xh = (bf16_to_f32(y[p * ldy + c]) - mu) * is;
    dy[p * lddy + c] = f32_to_bf16(k1 * (g - s / n - xh * sx / n));
  }
}

// ------------------------------------------------------------------ weight re-layout
// src: fp32 OIHW tensor with arbitrary element strides; dst_fwd [Opad][taps][Ipad] bf16 and
// (optionally) dst_t [Ipad][taps][Opad] bf16 (the dgrad operand).  Padding rows/cols are zero.
__global__ void k_prep_weights(const float* __restrict__ w, long long so, long long si, long long sh, long long sw, int O,
                               int I, int KH, int KW, int Opad, int Ipad, bf16_t* __restrict__ dst_fwd,
                               bf16_t* __restrict__ dst_t) {
  const int taps = KH * KW;
  const long long total = (long long)Opad * taps * Ipad;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(idx % Ipad);
    long long t = idx / Ipad;
    const int tap = (int)(t % taps);
    const int o = (int)(t / taps);
    float v = 0.f;
    if (o < O && i < I) v = w[o * so + i * si + (tap / KW) * sh + (tap % KW) * sw];
    const bf16_t b = f32_to_bf16(v);
    dst_fwd[idx] = b;
    if (dst_t) dst_t[((size_t)i * taps + tap) * Opad + o] = b;
  }
}

// stem: fp32 [O][3][7][7] (any strides) -> bf16 [O][7][32], element s*4+c of filter row r
__global__ void k_prep_stem(const float* __restrict__ w, long long so, long long si, long long sh, long long sw, int O,
                            bf16_t* __restrict__ dst) {
  const int total = O * 7 * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx % 32, r = (idx / 32) % 7, o = idx / (32 * 7);
    const int s = e >> 2, c = e & 3;
    float v = 0.f;
    if (s < 7 && c < 3) v = w[o * so + c * si + r * sh + s * sw];
    dst[idx] = f32_to_bf16(v);
  }
}

// stem gradient back: fp32 [O][7][32] -> fp32 [O][3][7][7] with the parameter's strides
__global__ void k_unpack_stem_grad(const float* __restrict__ g, float* __restrict__ dw, long long so, long long si,
                                   long long sh, long long sw, int O) {
  const int total = O * 3 * 7 * 7;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int s = idx % 7, r = (idx / 7) % 7, c = (idx / 49) % 3, o = idx / 147;
    dw[o * so + c * si + r * sh + s * sw] = g[(o * 7 + r) * 32 + s * 4 + c];
  }
}

// ---- multi-tensor weight prep: one launch re-lays-out up to PREP_MAX convolution weights
constexpr int PREP_MAX = 40;
struct PrepTable {
  const float* w[PREP_MAX];
  bf16_t* dst_fwd[PREP_MAX];
  bf16_t* dst_t[PREP_MAX];
  long long so[PREP_MAX], si[PREP_MAX], sh[PREP_MAX], sw[PREP_MAX];
  int O[PREP_MAX], I[PREP_MAX], K[PREP_MAX], Opad[PREP_MAX], Ipad[PREP_MAX];
  int first_block[PREP_MAX + 1];
  int count;
};
// One workgroup re-lays-out a 64 (cout) x 64 (cin) tile of one tap: the transposed copy goes through LDS, so both
// destinations are written in 128-byte runs (a thread-per-element version scattered 2-byte stores Opad*taps apart
// for the transposed copy and took 0.68 ms per training step for ResNet-50's 25 M weights).
constexpr int PREP_TILE = 64;

__global__ void __launch_bounds__(256) k_prep_weights_multi(PrepTable t) {
  __shared__ bf16_t tile[PREP_TILE][PREP_TILE + 2];
  int ti = 0;
  while (ti + 1 < t.count && (int)blockIdx.x >= t.first_block[ti + 1]) ++ti;
  const int O = t.O[ti], I = t.I[ti], KW = t.K[ti], taps = KW * KW, Opad = t.Opad[ti], Ipad = t.Ipad[ti];
  const int tilesI = (Ipad + PREP_TILE - 1) / PREP_TILE, tilesO = (Opad + PREP_TILE - 1) / PREP_TILE;
  int lb = (int)blockIdx.x - t.first_block[ti];
  const int i0 = (lb % tilesI) * PREP_TILE; lb /= tilesI;
  const int o0 = (lb % tilesO) * PREP_TILE;
  const int tap = lb / tilesO;
  const float* __restrict__ w = t.w[ti];
  bf16_t* __restrict__ df = t.dst_fwd[ti];
  bf16_t* __restrict__ dt = t.dst_t[ti];
  const long long so = t.so[ti], si = t.si[ti];
  const long long tapoff = (tap / KW) * t.sh[ti] + (tap % KW) * t.sw[ti];
  const int col = threadIdx.x & 63, row0 = threadIdx.x >> 6;
#pragma unroll 4
  for (int e = 0; e < PREP_TILE / 4; ++e) {
    const int o = o0 + e * 4 + row0, i = i0 + col;
    float v = 0.f;
    if (o < O && i < I) v = w[o * so + i * si + tapoff];
    const bf16_t bv = f32_to_bf16(v);
    tile[e * 4 + row0][col] = bv;
    if (o < Opad && i < Ipad) df[((size_t)o * taps + tap) * Ipad + i] = bv;
  }
  if (!dt) return;
  __syncthreads();
#pragma unroll 4
  for (int e = 0; e < PREP_TILE / 4; ++e) {
    const int i = i0 + e * 4 + row0, o = o0 + col;
    if (i < Ipad && o < Opad) dt[((size_t)i * taps + tap) * Opad + o] = tile[col][e * 4 + row0];
  }
}

int floor_pow2(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }
// grid for the elementwise BatchNorm kernels: rounded down to a multiple of CH / gcd(CH, 256) so that the grid stride
// keeps every thread on one channel chunk (any C: DenseNet's 96, 160, ... too); *fixed says whether that holds
int ew_blocks(long long total);
// The streaming BatchNorm apply / backward-apply kernels (16 bytes per lane and tensor, a handful of FMAs) reach the HBM rate
// with four 256-thread workgroups per CU looping over the tensor; launching one workgroup per 256 chunks (16384 of them)
// was 6-17 % slower on the same tensors (tools/bench_bn.py: 1024@28 backward-apply 93 -> 78 us, 256@112 apply 237 -> 222 us)
// and 2.4 % in the ResNet-50 step -- the dispatcher hands out 60 rounds of short workgroups instead of one resident set.
int bn_stream_blocks(long long total) {
  static long long cap = [] { const char* e = getenv("YV1_BN_EW_BLOCKS"); long long v = e ? atoll(e) : 1024; return v < 64 ? 1024 : v; }();
  long long b = (total + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}
int fixed_chunk_grid(long long total, int CH, bool* fixed) {
  int blocks = bn_stream_blocks(total);
  int g = CH, b = 256;
  while (b) { const int t = g % b; g = b; b = t; }      // gcd(CH, 256)
  const int m = CH / g;
  if (blocks >= m) blocks -= blocks % m;
  *fixed = ((long long)blocks * 256) % CH == 0;
  return blocks;
}
int ew_blocks(long long total) { long long b = (total + 255) / 256; return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }

}  // namespace

// ---- partial-row helpers ---------------------------------------------------------------------
// Sums groups of RB rows of a [rows][W] fp32 matrix: out[ceil(rows/RB)][W].
extern "C" int yv1_reduce_rows(const float* in, float* out, int rows, int W, int RB, hipStream_t stream) {
  if (!in || !out || rows <= 0 || W <= 0 || RB <= 0) return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_reduce_rows, dim3((W + 127) / 128, (rows + RB - 1) / RB), dim3(128), 0, stream, in, out, rows, W, RB);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// Folds partial statistic rows [rows][2][Cseg] into row form inside table [2][ldo] at channel c0
// (DenseNet: one statistics table per dense block, filled as each 32-channel slice is produced).
extern "C" int yv1_stats_merge(const float* partials, int rows, int Cseg, float* table, int ldo, int c0, hipStream_t stream) {
  if (!partials || !table || rows <= 0 || Cseg <= 0 || c0 < 0 || c0 + Cseg > ldo) return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_stats_merge, dim3((2 * Cseg + 63) / 64), dim3(1024), 0, stream, partials, rows, Cseg, table, ldo, c0);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// channels per workgroup of the finalize kernels (tuning: YV1_BN_FIN_CPB=8|16|64, YV1_BN_FIN_MAXC=<widest C split>)
static int finalize_cpb(int C) {
  static int cpb = 0, maxc = 0;
  if (!cpb) {
    const char* e = getenv("YV1_BN_FIN_CPB"); cpb = e ? atoi(e) : 16;
    if (cpb != 8 && cpb != 16 && cpb != 64) cpb = 16;
    const char* m = getenv("YV1_BN_FIN_MAXC"); maxc = m ? atoi(m) : 4096;
  }
  return C <= maxc ? cpb : 64;
}

static int bn_finalize(const float* partials, int rows, int C, int ld_partials, float count, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                       float* invstd, float* scale, float* shift, const float* seg_part, int seg_rows, int seg_c0, int seg_c,
                       float* table_out, hipStream_t stream) {
  if (!partials || rows <= 0 || C <= 0 || !mean || !invstd || !scale || !shift) return YV1_ERR_BAD_ARG;
  const int cpb = finalize_cpb(C);
  if (cpb == 8)
    hipLaunchKernelGGL(k_bn_finalize<8>, dim3((C + 7) / 8), dim3(1024), 0, stream, partials, rows, C, ld_partials, count, gamma,
                       beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, seg_part, seg_rows, seg_c0,
                       seg_c, table_out);
  else if (cpb == 16)
    hipLaunchKernelGGL(k_bn_finalize<16>, dim3((C + 15) / 16), dim3(1024), 0, stream, partials, rows, C, ld_partials, count, gamma,
                       beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, seg_part, seg_rows, seg_c0,
                       seg_c, table_out);
  else
    hipLaunchKernelGGL(k_bn_finalize<64>, dim3((C + 63) / 64), dim3(1024), 0, stream, partials, rows, C, ld_partials, count, gamma,
                       beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, seg_part, seg_rows, seg_c0,
                       seg_c, table_out);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_bn_finalize(const float* partials, int rows, int C, int ld_partials, float count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               float* mean, float* invstd, float* scale, float* shift, hipStream_t stream) {
  return bn_finalize(partials, rows, C, ld_partials, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                     scale, shift, nullptr, 0, 0, 0, nullptr, stream);
}

// DenseNet (OriginDenseNet.py:32-36: every layer's norm1 acts on the concatenation of all earlier features): the shared
// one-row table [2][ld] of channel sums / sums of squares is finalized for the first C channels, of which
// [seg_c0, seg_c0 + seg_c) -- the features the previous layer just produced -- are still that convolution's partial rows
// seg_part [seg_rows][2][seg_c]: they are summed here AND written into the table (what yv1_stats_merge + yv1_bn_finalize
// do in two launches).  seg_c0 + seg_c <= C.
extern "C" int yv1_bn_finalize_merged(float* table, int C, int ld, float count, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var, float* mean,
                                      float* invstd, float* scale, float* shift, const float* seg_part, int seg_rows,
                                      int seg_c0, int seg_c, hipStream_t stream) {
  if (!seg_part || seg_rows <= 0 || seg_c <= 0 || seg_c0 < 0 || seg_c0 + seg_c > C || C > ld) return YV1_ERR_BAD_ARG;
  return bn_finalize(table, 1, C, ld, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift,
                     seg_part, seg_rows, seg_c0, seg_c, table, stream);
}

extern "C" int yv1_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift, hipStream_t stream) {
  if (C <= 0 || !gamma || !beta || !running_mean || !running_var || !scale || !shift) return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_bn_eval_coeffs, dim3((C + 63) / 64), dim3(64), 0, stream, C, gamma, beta, running_mean, running_var,
                     eps, scale, shift);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

static int bn_apply_launch(const void* y, int ldy, void* z, int ldz, const void* residual, int ldr, const float* scale,
                           const float* shift, const float* res_scale, const float* res_shift, long long npix, int C,
                           int relu, void* relu_mask, void* z8, int ldz8, hipStream_t stream) {
  if (!y || !z || !scale || !shift || npix <= 0 || C <= 0) return YV1_ERR_BAD_ARG;
  if (C % 8 || ldy % 8 || ldz % 8 || (residual && ldr % 8) || (z8 && ldz8 % 8)) return YV1_ERR_UNSUPPORTED;
  ApplyArgs a;
  a.y = (const bf16_t*)y; a.ldy = ldy; a.z = (bf16_t*)z; a.ldz = ldz; a.res = (const bf16_t*)residual; a.ldr = ldr;
  a.scale = scale; a.shift = shift; a.rscale = res_scale; a.rshift = res_shift; a.npix = npix; a.C = C; a.relu = relu;
  a.relu_mask = (unsigned char*)relu_mask; a.z8 = (unsigned char*)z8; a.ldz8 = ldz8;
  bool fixed;
  const int blocks = fixed_chunk_grid(npix * (C / 8), C / 8, &fixed);
  if (fixed) hipLaunchKernelGGL(k_bn_apply<true>, dim3(blocks), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(k_bn_apply<false>, dim3(blocks), dim3(256), 0, stream, a);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_bn_apply(const void* y, int ldy, void* z, int ldz, const void* residual, int ldr, const float* scale,
                            const float* shift, const float* res_scale, const float* res_shift, long long npix, int C,
                            int relu, void* relu_mask, hipStream_t stream) {
  return bn_apply_launch(y, ldy, z, ldz, residual, ldr, scale, shift, res_scale, res_shift, npix, C, relu, relu_mask, nullptr,
                         0, stream);
}

// yv1_bn_apply with a second, e4m3 copy of the result (pixel stride ldz8 bytes): the operand of the next fp8 convolution
extern "C" int yv1_bn_apply_q8(const void* y, int ldy, void* z, int ldz, const void* residual, int ldr, const float* scale,
                               const float* shift, const float* res_scale, const float* res_shift, long long npix, int C,
                               int relu, void* relu_mask, void* z8, int ldz8, hipStream_t stream) {
  if (!z8) return YV1_ERR_BAD_ARG;
  return bn_apply_launch(y, ldy, z, ldz, residual, ldr, scale, shift, res_scale, res_shift, npix, C, relu, relu_mask, z8,
                         ldz8, stream);
}

// ---- finalize + apply in one launch (see "finalize fused into the consuming launch" above) ----------------------------
// `sync`: two zero-initialised unsigned words owned by this launch until it completes (the kernel leaves them zero again);
// `fault`: one word the host checks -- a consumer that gave up waiting sets it (never observed).
static FinFwd make_fin_fwd(const float* partials, int rows, int C, int ld, float count, const float* gamma, const float* beta,
                           float eps, float momentum, float* rmean, float* rvar, float* mean, float* invstd, float* scale,
                           float* shift) {
  FinFwd f;
  f.part = partials; f.rows = rows; f.C = C; f.ldp = ld; f.count = count; f.gamma = gamma; f.beta = beta; f.eps = eps;
  f.momentum = momentum; f.rmean = rmean; f.rvar = rvar; f.mean = mean; f.invstd = invstd; f.scale = scale; f.shift = shift;
  f.nblk = partials ? (C + FCPB - 1) / FCPB : 0;
  return f;
}

extern "C" int yv1_bn_finalize_apply(const float* partials, int rows, int ld_partials, float count, const float* gamma,
                                     const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                     float* mean, float* invstd, float* scale, float* shift, const float* r_partials,
                                     int r_rows, int r_ld_partials, const float* r_gamma, const float* r_beta,
                                     float* r_running_mean, float* r_running_var, float* r_mean, float* r_invstd,
                                     float* r_scale, float* r_shift, const void* y, int ldy, void* z, int ldz,
                                     const void* residual, int ldr, long long npix, int C, int relu, void* relu_mask,
                                     void* z8, int ldz8, unsigned* sync, unsigned* fault, hipStream_t stream) {
  if (!partials || rows <= 0 || !mean || !invstd || !scale || !shift || !y || !z || npix <= 0 || C <= 0 || !sync || !fault)
    return YV1_ERR_BAD_ARG;
  if (r_partials && (r_rows <= 0 || !r_mean || !r_invstd || !r_scale || !r_shift || !residual)) return YV1_ERR_BAD_ARG;
  if (C % 8 || C > 4096 || ldy % 8 || ldz % 8 || (residual && ldr % 8) || (z8 && ldz8 % 8)) return YV1_ERR_UNSUPPORTED;
  ApplyArgs a;
  a.y = (const bf16_t*)y; a.ldy = ldy; a.z = (bf16_t*)z; a.ldz = ldz; a.res = (const bf16_t*)residual; a.ldr = ldr;
  a.scale = scale; a.shift = shift; a.rscale = r_scale; a.rshift = r_shift; a.npix = npix; a.C = C; a.relu = relu;
  a.relu_mask = (unsigned char*)relu_mask; a.z8 = (unsigned char*)z8; a.ldz8 = ldz8;
  const FinFwd f1 = make_fin_fwd(partials, rows, C, ld_partials, count, gamma, beta, eps, momentum, running_mean, running_var,
                                 mean, invstd, scale, shift);
  const FinFwd f2 = make_fin_fwd(r_partials, r_rows, C, r_ld_partials, count, r_gamma, r_beta, eps, momentum, r_running_mean,
                                 r_running_var, r_mean, r_invstd, r_scale, r_shift);
  FuseSync s; s.ctr = sync; s.fault = fault; s.producers = f1.nblk + f2.nblk;
  bool fixed;
  const int blocks = fixed_chunk_grid(npix * (C / 8), C / 8, &fixed);
  const size_t lds = sizeof(float) * (size_t)((4 * C) > 2 * FRL * FCPB ? 4 * C : 2 * FRL * FCPB);
  if (fixed) hipLaunchKernelGGL(k_bn_apply_fused<true>, dim3(s.producers + blocks), dim3(256), lds, stream, a, f1, f2, s);
  else hipLaunchKernelGGL(k_bn_apply_fused<false>, dim3(s.producers + blocks), dim3(256), lds, stream, a, f1, f2, s);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// DenseNet form of yv1_bn_finalize_apply (see yv1_bn_finalize_merged): `table` is the block's one-row [2][ld] table of
// channel sums; its columns [seg_c0, seg_c0 + seg_c) are still seg_part [seg_rows][2][seg_c] (seg_part may be NULL: plain
// table); the BatchNorm acts on the first C channels of y (a window of the block buffer), no residual.
extern "C" int yv1_bn_finalize_merged_apply(float* table, int C, int ld, float count, const float* gamma, const float* beta,
                                            float eps, float momentum, float* running_mean, float* running_var, float* mean,
                                            float* invstd, float* scale, float* shift, const float* seg_part, int seg_rows,
                                            int seg_c0, int seg_c, const void* y, int ldy, void* z, int ldz, long long npix,
                                            int relu, unsigned* sync, unsigned* fault, hipStream_t stream) {
  if (!table || C <= 0 || C > ld || !mean || !invstd || !scale || !shift || !y || !z || npix <= 0 || !sync || !fault)
    return YV1_ERR_BAD_ARG;
  if (seg_part && (seg_rows <= 0 || seg_c <= 0 || seg_c0 < 0 || seg_c0 + seg_c > C)) return YV1_ERR_BAD_ARG;
  if (C % 8 || C > 4096 || ldy % 8 || ldz % 8) return YV1_ERR_UNSUPPORTED;
  ApplyArgs a;
  a.y = (const bf16_t*)y; a.ldy = ldy; a.z = (bf16_t*)z; a.ldz = ldz; a.res = nullptr; a.ldr = 0;
  a.scale = scale; a.shift = shift; a.rscale = nullptr; a.rshift = nullptr; a.npix = npix; a.C = C; a.relu = relu;
  a.relu_mask = nullptr; a.z8 = nullptr; a.ldz8 = 0;
  FinFwd f1 = make_fin_fwd(table, 1, C, ld, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale,
                           shift);
  f1.seg_part = seg_part; f1.seg_rows = seg_rows; f1.seg_c0 = seg_c0; f1.seg_c = seg_c; f1.table_out = table;
  const FinFwd f2 = make_fin_fwd(nullptr, 0, C, 0, count, nullptr, nullptr, eps, momentum, nullptr, nullptr, nullptr, nullptr,
                                 nullptr, nullptr);
  FuseSync s; s.ctr = sync; s.fault = fault; s.producers = f1.nblk;
  bool fixed;
  const int blocks = fixed_chunk_grid(npix * (C / 8), C / 8, &fixed);
  const size_t lds = sizeof(float) * (size_t)((4 * C) > 2 * FRL * FCPB ? 4 * C : 2 * FRL * FCPB);
  if (fixed) hipLaunchKernelGGL(k_bn_apply_fused<true>, dim3(s.producers + blocks), dim3(256), lds, stream, a, f1, f2, s);
  else hipLaunchKernelGGL(k_bn_apply_fused<false>, dim3(s.producers + blocks), dim3(256), lds, stream, a, f1, f2, s);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

static FinBwd make_fin_bwd(const float* partials, int rows, int C, float count, const float* gamma, const float* invstd,
                           float* dgamma, float* dbeta, float* k1, float* k2, float* k3) {
  FinBwd f;
  f.part = partials; f.rows = rows; f.C = C; f.count = count; f.gamma = gamma; f.invstd = invstd; f.dgamma = dgamma;
  f.dbeta = dbeta; f.k1 = k1; f.k2 = k2; f.k3 = k3; f.nblk = partials ? (C + FCPB - 1) / FCPB : 0;
  return f;
}

// yv1_bn_bwd_finalize + yv1_bn_bwd_apply in one launch (partials from yv1_bn_bwd_reduce); k1/k2/k3 are scratch [C] each.
extern "C" int yv1_bn_bwd_finalize_apply(const float* partials, int rows, float count, const float* gamma, float* dgamma,
                                         float* dbeta, float* k1, float* k2, float* k3, const void* dz, int lddz,
                                         const void* z, int ldz, const void* y, int ldy, const float* mean,
                                         const float* invstd, const float* scale, const float* shift, long long npix, int C,
                                         int mask_mode, void* dy, int lddy, void* dres, int lddres, int accumulate,
                                         unsigned* sync, unsigned* fault, hipStream_t stream) {
  if (!partials || rows <= 0 || !dz || !y || !mean || !invstd || !k1 || !k2 || !k3 || !dy || npix <= 0 || !sync || !fault)
    return YV1_ERR_BAD_ARG;
  if (((mask_mode == 1 || mask_mode == 3) && !z) || (mask_mode == 2 && (!scale || !shift))) return YV1_ERR_BAD_ARG;
  if (C % 8 || C > 4096 || lddz % 8 || ldy % 8 || lddy % 8 || (z && mask_mode == 1 && ldz % 8) || (dres && lddres % 8))
    return YV1_ERR_UNSUPPORTED;
  BwdArgs a = {};
  a.dz = (const bf16_t*)dz; a.lddz = lddz; a.z = (const bf16_t*)z; a.ldz = ldz; a.y = (const bf16_t*)y; a.ldy = ldy;
  a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.npix = npix; a.C = C; a.mask_mode = mask_mode;
  a.k1 = k1; a.k2 = k2; a.k3 = k3; a.dy = (bf16_t*)dy; a.lddy = lddy; a.dres = (bf16_t*)dres; a.lddres = lddres;
  a.accumulate = accumulate;
  const FinBwd f1 = make_fin_bwd(partials, rows, C, count, gamma, invstd, dgamma, dbeta, k1, k2, k3);
  FuseSync s; s.ctr = sync; s.fault = fault; s.producers = f1.nblk;
  bool fixed;
  const int blocks = fixed_chunk_grid(npix * (C / 8), C / 8, &fixed);
  const size_t lds = sizeof(float) * (size_t)((3 * C) > 2 * FRL * FCPB ? 3 * C : 2 * FRL * FCPB);
  if (fixed) hipLaunchKernelGGL(k_bn_bwd_apply_fused<true>, dim3(s.producers + blocks), dim3(256), lds, stream, a, f1, s);
  else hipLaunchKernelGGL(k_bn_bwd_apply_fused<false>, dim3(s.producers + blocks), dim3(256), lds, stream, a, f1, s);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// Dual form: both finalizes (partials / partials2 from yv1_bn_bwd_reduce_dual) + yv1_bn_bwd_apply_dual in one launch.
extern "C" int yv1_bn_bwd_finalize_apply_dual(const float* partials, const float* partials2, int rows, float count,
                                              const float* gamma, const float* gamma2, float* dgamma, float* dbeta,
                                              float* dgamma2, float* dbeta2, float* k6 /* [6][C] scratch */, const void* dz,
                                              int lddz, const void* z, int ldz, const void* y, int ldy, const float* mean,
                                              const float* invstd, void* dy, int lddy, const void* y2, int ldy2,
                                              const float* mean2, const float* invstd2, void* dy2, int lddy2, long long npix,
                                              int C, int mask_mode, unsigned* sync, unsigned* fault, hipStream_t stream) {
  if (!partials || !partials2 || rows <= 0 || !k6 || !dz || !y || !y2 || !mean || !invstd || !mean2 || !invstd2 || !dy || !dy2 ||
      npix <= 0 || !sync || !fault)
    return YV1_ERR_BAD_ARG;
  if (mask_mode == 2 || ((mask_mode == 1 || mask_mode == 3) && !z)) return YV1_ERR_BAD_ARG;
  if (C % 8 || C > 2048 || lddz % 8 || ldy % 8 || ldy2 % 8 || lddy % 8 || lddy2 % 8 || (z && mask_mode == 1 && ldz % 8))
    return YV1_ERR_UNSUPPORTED;
  BwdArgs a = {};
  a.dz = (const bf16_t*)dz; a.lddz = lddz; a.z = (const bf16_t*)z; a.ldz = ldz; a.y = (const bf16_t*)y; a.ldy = ldy;
  a.mean = mean; a.invstd = invstd; a.npix = npix; a.C = C; a.mask_mode = mask_mode;
  a.k1 = k6; a.k2 = k6 + C; a.k3 = k6 + 2 * C; a.dy = (bf16_t*)dy; a.lddy = lddy;
  a.y2 = (const bf16_t*)y2; a.ldy2 = ldy2; a.mean2 = mean2; a.invstd2 = invstd2; a.k1b = k6 + 3 * C; a.k2b = k6 + 4 * C;
  a.k3b = k6 + 5 * C; a.dy2 = (bf16_t*)dy2; a.lddy2 = lddy2;
  const FinBwd f1 = make_fin_bwd(partials, rows, C, count, gamma, invstd, dgamma, dbeta, k6, k6 + C, k6 + 2 * C);
  const FinBwd f2 = make_fin_bwd(partials2, rows, C, count, gamma2, invstd2, dgamma2, dbeta2, k6 + 3 * C, k6 + 4 * C, k6 + 5 * C);
  FuseSync s; s.ctr = sync; s.fault = fault; s.producers = f1.nblk + f2.nblk;
  bool fixed;
  const int blocks = fixed_chunk_grid(npix * (C / 8), C / 8, &fixed);
  if (!fixed) return YV1_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * (size_t)((6 * C) > 2 * FRL * FCPB ? 6 * C : 2 * FRL * FCPB);
  hipLaunchKernelGGL(k_bn_bwd_apply_dual_fused, dim3(s.producers + blocks), dim3(256), lds, stream, a, f1, f2, s);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// pixels per workgroup of the column-parallel reductions: aim for ~2048 workgroups (8 per CU) so small
// feature maps are not a few long serial chains; at least 4 pixel iterations per thread
static long long reduce_ppb(long long npix, int TY) {
  static const long long target = [] { const char* e = getenv("YV1_BN_BLOCKS"); long long v = e ? atoll(e) : 1024; return v < 64 ? 1024 : v; }();
  long long p = (npix + target - 1) / target;
  p = (p + TY - 1) / TY * TY;
  if (p < (long long)TY * 4) p = (long long)TY * 4;
  return p;
}

// number of partial rows yv1_bn_stats / yv1_bn_bwd_reduce write for npix pixels
extern "C" int yv1_bn_reduce_rows(long long npix, int C) {
  const int CH = C / 8;
  const int TX = floor_pow2(CH < 256 ? CH : 256), TY = 256 / TX;
  return (int)((npix + reduce_ppb(npix, TY) - 1) / reduce_ppb(npix, TY));
}

static int reduce_geometry(long long npix, int C, int* TX, int* ppb, int* blocks, size_t* lds) {
  if (C % 8 || C > 4096) return YV1_ERR_UNSUPPORTED;
  const int CH = C / 8;
  *TX = floor_pow2(CH < 256 ? CH : 256);
  if (CH >= 2 * *TX) return YV1_ERR_UNSUPPORTED;
  const int TY = 256 / *TX;
  const long long p = reduce_ppb(npix, TY);
  *ppb = (int)p; *blocks = (int)((npix + p - 1) / p);
  *lds = (size_t)TY * 2 * C * sizeof(float);
  return YV1_OK;
}

// stand-alone batch statistics: partials [yv1_bn_reduce_rows(npix,C)][2][C]
extern "C" int yv1_bn_stats(const void* y, int ldy, long long npix, int C, float* partials, hipStream_t stream) {
  if (!y || !partials || npix <= 0) return YV1_ERR_BAD_ARG;
  int TX, ppb, blocks; size_t lds;
  int rc = reduce_geometry(npix, C, &TX, &ppb, &blocks, &lds);
  if (rc) return rc;
  if (ldy % 8) return YV1_ERR_UNSUPPORTED;
  StatsArgs a; a.y = (const bf16_t*)y; a.ldy = ldy; a.npix = npix; a.C = C; a.pix_per_block = ppb; a.part = partials;
  if (C / 8 > TX) hipLaunchKernelGGL(k_bn_stats<2>, dim3(blocks), dim3(256), lds, stream, a, TX);
  else hipLaunchKernelGGL(k_bn_stats<1>, dim3(blocks), dim3(256), lds, stream, a, TX);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

static int bn_bwd_reduce(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy, const float* mean,
                         const float* invstd, const float* scale, const float* shift, long long npix, int C,
                         int mask_mode, float* partials, const void* pool_idx, int pH, int pW, hipStream_t stream) {
  if (!dz || !y || !mean || !invstd || !partials || npix <= 0) return YV1_ERR_BAD_ARG;
  if (pool_idx && (pH <= 0 || pW <= 0 || npix % ((long long)pH * pW) || npix >= (1ll << 31))) return YV1_ERR_BAD_ARG;
  if (((mask_mode == 1 || mask_mode == 3) && !z) || (mask_mode == 2 && (!scale || !shift))) return YV1_ERR_BAD_ARG;
  int TX, ppb, blocks; size_t lds;
  int rc = reduce_geometry(npix, C, &TX, &ppb, &blocks, &lds);
  if (rc) return rc;
  if (lddz % 8 || ldy % 8 || (z && mask_mode == 1 && ldz % 8)) return YV1_ERR_UNSUPPORTED;
  BwdArgs a = {};
  a.dz = (const bf16_t*)dz; a.lddz = lddz; a.z = (const bf16_t*)z; a.ldz = ldz; a.y = (const bf16_t*)y; a.ldy = ldy;
  a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.npix = npix; a.C = C; a.pix_per_block = ppb;
  a.mask_mode = mask_mode; a.part = partials;
  a.pool_idx = (const unsigned char*)pool_idx; a.pH = pH; a.pW = pW;
  if (pool_idx && pH % 2 == 0 && pW % 2 == 0 && C / 8 <= TX) {       // aligned 2x2 patches: same partial rows, patch ranges
    a.npix = npix / 4;
    a.pix_per_block = (int)((a.npix + blocks - 1) / blocks);
    hipLaunchKernelGGL(k_bn_bwd_reduce_pool2, dim3(blocks), dim3(256), lds, stream, a, TX);
    YV1_LAUNCH_CHECK();
    return YV1_OK;
  }
  if (!pool_idx) {
#define YV1_BWD_REDUCE_S(MODE_)                                                                                 \
    if (mask_mode == MODE_) {                                                                                   \
      if (C / 8 > TX) hipLaunchKernelGGL((k_bn_bwd_reduce_s<2, MODE_>), dim3(blocks), dim3(256), lds, stream, a, TX);  \
      else hipLaunchKernelGGL((k_bn_bwd_reduce_s<1, MODE_>), dim3(blocks), dim3(256), lds, stream, a, TX);      \
    }
    YV1_BWD_REDUCE_S(0) YV1_BWD_REDUCE_S(1) YV1_BWD_REDUCE_S(2) YV1_BWD_REDUCE_S(3)
#undef YV1_BWD_REDUCE_S
  }
  else if (C / 8 > TX) hipLaunchKernelGGL(k_bn_bwd_reduce<2>, dim3(blocks), dim3(256), lds, stream, a, TX);
  else hipLaunchKernelGGL(k_bn_bwd_reduce<1>, dim3(blocks), dim3(256), lds, stream, a, TX);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_bn_bwd_reduce(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, long long npix, int C,
                                 int mask_mode, float* partials, hipStream_t stream) {
  return bn_bwd_reduce(dz, lddz, z, ldz, y, ldy, mean, invstd, scale, shift, npix, C, mask_mode, partials, nullptr, 0, 0, stream);
}

// BatchNorm(+ReLU) backward behind a 3x3/2 max pool (the stems, OriginResNet.py:174-177 / OriginDenseNet.py:120-128):
// dpool [N,OH,OW,C] is the gradient of the POOL OUTPUT; the pool's backward is gathered on the fly from pool_idx, so the
// gradient of the pool input ([N,H,W,C], 4x the bytes) is never written or re-read.  y is the BatchNorm input [N,H,W,C].
extern "C" int yv1_bn_bwd_reduce_pooled(const void* dpool, int lddp, const void* pool_idx, const void* y, int ldy,
                                        const float* mean, const float* invstd, const float* scale, const float* shift,
                                        int N, int H, int W, int C, int mask_mode, float* partials, hipStream_t stream) {
  if (!pool_idx || N <= 0 || (mask_mode != 0 && mask_mode != 2)) return YV1_ERR_BAD_ARG;
  return bn_bwd_reduce(dpool, lddp, nullptr, 0, y, ldy, mean, invstd, scale, shift, (long long)N * H * W, C, mask_mode, partials,
                       pool_idx, H, W, stream);
}

extern "C" int yv1_bn_bwd_finalize(const float* partials, int rows, int C, float count, const float* gamma,
                                   const float* invstd, float* dgamma, float* dbeta, float* k1, float* k2, float* k3,
                                   hipStream_t stream) {
  if (!partials || rows <= 0 || C <= 0 || !invstd || !k1 || !k2 || !k3) return YV1_ERR_BAD_ARG;
  const int cpb = finalize_cpb(C);
  if (cpb == 8)
    hipLaunchKernelGGL(k_bn_bwd_finalize<8>, dim3((C + 7) / 8), dim3(1024), 0, stream, partials, rows, C, count, gamma,
                       invstd, dgamma, dbeta, k1, k2, k3);
  else if (cpb == 16)
    hipLaunchKernelGGL(k_bn_bwd_finalize<16>, dim3((C + 15) / 16), dim3(1024), 0, stream, partials, rows, C, count, gamma,
                       invstd, dgamma, dbeta, k1, k2, k3);
  else
    hipLaunchKernelGGL(k_bn_bwd_finalize<64>, dim3((C + 63) / 64), dim3(1024), 0, stream, partials, rows, C, count, gamma,
                       invstd, dgamma, dbeta, k1, k2, k3);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

static int bn_bwd_apply(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy, const float* mean,
                        const float* invstd, const float* scale, const float* shift, const float* k1,
                        const float* k2, const float* k3, long long npix, int C, int mask_mode, void* dy, int lddy,
                        void* dres, int lddres, int accumulate, const void* pool_idx, int pH, int pW, hipStream_t stream) {
  if (!dz || !y || !mean || !invstd || !k1 || !k2 || !k3 || !dy || npix <= 0) return YV1_ERR_BAD_ARG;
  if (pool_idx && (pH <= 0 || pW <= 0 || npix % ((long long)pH * pW) || npix >= (1ll << 31))) return YV1_ERR_BAD_ARG;
  if (((mask_mode == 1 || mask_mode == 3) && !z) || (mask_mode == 2 && (!scale || !shift))) return YV1_ERR_BAD_ARG;
  if (C % 8 || lddz % 8 || ldy % 8 || lddy % 8 || (z && mask_mode == 1 && ldz % 8) || (dres && lddres % 8)) return YV1_ERR_UNSUPPORTED;
  BwdArgs a = {};
  a.dz = (const bf16_t*)dz; a.lddz = lddz; a.z = (const bf16_t*)z; a.ldz = ldz; a.y = (const bf16_t*)y; a.ldy = ldy;
  a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.npix = npix; a.C = C; a.mask_mode = mask_mode;
  a.k1 = k1; a.k2 = k2; a.k3 = k3; a.dy = (bf16_t*)dy; a.lddy = lddy; a.dres = (bf16_t*)dres; a.lddres = lddres;
  a.accumulate = accumulate;
  a.pool_idx = (const unsigned char*)pool_idx; a.pH = pH; a.pW = pW;
  if (pool_idx && pH % 2 == 0 && pW % 2 == 0 && (npix / 4) * (C / 8) < (1ll << 32)) {   // 32-bit patch index arithmetic
    a.npix = npix / 4;
    hipLaunchKernelGGL(k_bn_bwd_apply_pool2, dim3(ew_blocks(a.npix * (C / 8))), dim3(256), 0, stream, a);
    YV1_LAUNCH_CHECK();
    return YV1_OK;
  }
  bool fixed;
  const int blocks = fixed_chunk_grid(npix * (C / 8), C / 8, &fixed);
  if (fixed) hipLaunchKernelGGL(k_bn_bwd_apply<true>, dim3(blocks), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(k_bn_bwd_apply<false>, dim3(blocks), dim3(256), 0, stream, a);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_bn_bwd_apply(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy, const float* mean,
                                const float* invstd, const float* scale, const float* shift, const float* k1,
                                const float* k2, const float* k3, long long npix, int C, int mask_mode, void* dy, int lddy,
                                void* dres, int lddres, int accumulate, hipStream_t stream) {
  return bn_bwd_apply(dz, lddz, z, ldz, y, ldy, mean, invstd, scale, shift, k1, k2, k3, npix, C, mask_mode, dy, lddy, dres, lddres,
                      accumulate, nullptr, 0, 0, stream);
}

extern "C" int yv1_bn_bwd_apply_pooled(const void* dpool, int lddp, const void* pool_idx, const void* y, int ldy,
                                       const float* mean, const float* invstd, const float* scale, const float* shift,
                                       const float* k1, const float* k2, const float* k3, int N, int H, int W, int C,
                                       int mask_mode, void* dy, int lddy, hipStream_t stream) {
  if (!pool_idx || N <= 0 || (mask_mode != 0 && mask_mode != 2)) return YV1_ERR_BAD_ARG;
  return bn_bwd_apply(dpool, lddp, nullptr, 0, y, ldy, mean, invstd, scale, shift, k1, k2, k3, (long long)N * H * W, C, mask_mode,
                      dy, lddy, nullptr, 0, 0, pool_idx, H, W, stream);
}

// Dual BatchNorm backward (projection Bottleneck: bn3 and the downsample BatchNorm receive the same ReLU-masked
// gradient, OriginResNet.py:100-105): ONE reduction pass and ONE apply pass read dz and the mask for both.
// mask_mode 0 / 1 / 3 as yv1_bn_bwd_reduce (2 -- mask from one BatchNorm's own output -- makes no sense for a sum).
extern "C" int yv1_bn_bwd_reduce_dual(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy,
                                      const float* mean, const float* invstd, const void* y2, int ldy2, const float* mean2,
                                      const float* invstd2, long long npix, int C, int mask_mode, float* partials,
                                      float* partials2, hipStream_t stream) {
  if (!dz || !y || !y2 || !mean || !invstd || !mean2 || !invstd2 || !partials || !partials2 || npix <= 0) return YV1_ERR_BAD_ARG;
  if (mask_mode == 2 || ((mask_mode == 1 || mask_mode == 3) && !z)) return YV1_ERR_BAD_ARG;
  int TX, ppb, blocks; size_t lds;
  int rc = reduce_geometry(npix, C, &TX, &ppb, &blocks, &lds);
  if (rc) return rc;
  if (lddz % 8 || ldy % 8 || ldy2 % 8 || (z && mask_mode == 1 && ldz % 8)) return YV1_ERR_UNSUPPORTED;
  BwdArgs a = {};
  a.dz = (const bf16_t*)dz; a.lddz = lddz; a.z = (const bf16_t*)z; a.ldz = ldz; a.y = (const bf16_t*)y; a.ldy = ldy;
  a.mean = mean; a.invstd = invstd; a.npix = npix; a.C = C; a.pix_per_block = ppb; a.mask_mode = mask_mode; a.part = partials;
  a.y2 = (const bf16_t*)y2; a.ldy2 = ldy2; a.mean2 = mean2; a.invstd2 = invstd2; a.part2 = partials2;
#define YV1_REDUCE_DUAL(MODE_)                                                                                   \
  if (mask_mode == MODE_) {                                                                                     \
    if (C / 8 > TX) hipLaunchKernelGGL((k_bn_bwd_reduce_dual<2, MODE_>), dim3(blocks), dim3(256), lds, stream, a, TX);  \
    else hipLaunchKernelGGL((k_bn_bwd_reduce_dual<1, MODE_>), dim3(blocks), dim3(256), lds, stream, a, TX);     \
  }
  YV1_REDUCE_DUAL(0) YV1_REDUCE_DUAL(1) YV1_REDUCE_DUAL(3)
#undef YV1_REDUCE_DUAL
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_bn_bwd_apply_dual(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy,
                                     const float* mean, const float* invstd, const float* k1, const float* k2,
                                     const float* k3, void* dy, int lddy, const void* y2, int ldy2, const float* mean2,
                                     const float* invstd2, const float* k1b, const float* k2b, const float* k3b, void* dy2,
                                     int lddy2, long long npix, int C, int mask_mode, hipStream_t stream) {
  if (!dz || !y || !y2 || !mean || !invstd || !mean2 || !invstd2 || !k1 || !k2 || !k3 || !k1b || !k2b || !k3b || !dy || !dy2 ||
      npix <= 0)
    return YV1_ERR_BAD_ARG;
  if (mask_mode == 2 || ((mask_mode == 1 || mask_mode == 3) && !z)) return YV1_ERR_BAD_ARG;
  if (C % 8 || lddz % 8 || ldy % 8 || ldy2 % 8 || lddy % 8 || lddy2 % 8 || (z && mask_mode == 1 && ldz % 8)) return YV1_ERR_UNSUPPORTED;
  BwdArgs a = {};
  a.dz = (const bf16_t*)dz; a.lddz = lddz; a.z = (const bf16_t*)z; a.ldz = ldz; a.y = (const bf16_t*)y; a.ldy = ldy;
  a.mean = mean; a.invstd = invstd; a.npix = npix; a.C = C; a.mask_mode = mask_mode;
  a.k1 = k1; a.k2 = k2; a.k3 = k3; a.dy = (bf16_t*)dy; a.lddy = lddy;
  a.y2 = (const bf16_t*)y2; a.ldy2 = ldy2; a.mean2 = mean2; a.invstd2 = invstd2; a.k1b = k1b; a.k2b = k2b; a.k3b = k3b;
  a.dy2 = (bf16_t*)dy2; a.lddy2 = lddy2;
  bool fixed;
  const int blocks = fixed_chunk_grid(npix * (C / 8), C / 8, &fixed);
  if (!fixed) return YV1_ERR_UNSUPPORTED;               // every ResNet width qualifies (C/8 a power of two)
  hipLaunchKernelGGL(k_bn_bwd_apply_dual, dim3(blocks), dim3(256), 0, stream, a);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_subsample2_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C,
                                        hipStream_t stream) {
  if (!x || !y || N <= 0 || (H & 1) || (W & 1) || C % 8 || ldx % 8 || ldy % 8) return YV1_ERR_BAD_ARG;
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(k_subsample2, dim3(ew_blocks(total)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, N, H, W, C);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// idx (nullable): uint8 [N,OH,OW,C], window position (r*3+s) of the first maximum, consumed by the backward
extern "C" int yv1_maxpool3x3s2_fwd(const void* x, int ldx, void* y, int ldy, void* idx, int N, int H, int W, int C,
                                    hipStream_t stream) {
  if (!x || !y || N <= 0 || C % 8 || ldx % 8 || ldy % 8) return YV1_ERR_BAD_ARG;
  const long long total = (long long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (C / 8);
  hipLaunchKernelGGL(k_maxpool_fwd, dim3(ew_blocks(total)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy,
                     (unsigned char*)idx, N, H, W, C);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// Either idx (from the forward) or x (the forward input: the first maximum is re-derived) must be given.
extern "C" int yv1_maxpool3x3s2_bwd(const void* x, int ldx, const void* idx, const void* dy, int lddy, void* dx, int lddx, int N,
                                    int H, int W, int C, hipStream_t stream) {
  if ((!x && !idx) || !dy || !dx || N <= 0 || C % 8 || (x && ldx % 8) || lddy % 8 || lddx % 8) return YV1_ERR_BAD_ARG;
  const long long total = (long long)N * H * W * (C / 8);
  if (idx)
    hipLaunchKernelGGL(k_maxpool_bwd_idx, dim3(ew_blocks(total)), dim3(256), 0, stream, (const unsigned char*)idx,
                       (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, N, H, W, C);
  else
    hipLaunchKernelGGL(k_maxpool_bwd, dim3(ew_blocks(total)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy,
                       (bf16_t*)dx, lddx, N, H, W, C);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// pooled [N,OH,OW,*] = maxpool3x3s2(bf16(relu?(y*scale + shift))), idx as yv1_maxpool3x3s2_fwd: the same values as
// yv1_bn_apply followed by yv1_maxpool3x3s2_fwd, bit for bit, without the intermediate tensor
extern "C" int yv1_bn_act_maxpool3x3s2_fwd(const void* y, int ldy, const float* scale, const float* shift, int relu, void* pooled,
                                           int ldp, void* idx, int N, int H, int W, int C, hipStream_t stream) {
  if (!y || !scale || !shift || !pooled || N <= 0 || H <= 0 || W <= 0 || C <= 0) return YV1_ERR_BAD_ARG;
  if (C % 8 || ldy % 8 || ldp % 8) return YV1_ERR_UNSUPPORTED;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * OH * OW * (C / 8);
  if (total >= (1ll << 32)) return YV1_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_bn_act_maxpool_fwd, dim3(ew_blocks(total)), dim3(256), 0, stream, (const bf16_t*)y, ldy, scale, shift, relu,
                     (bf16_t*)pooled, ldp, (unsigned char*)idx, N, H, W, C);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_avgpool2_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, hipStream_t stream) {
  if (!x || !y || N <= 0 || C % 8 || ldx % 8 || ldy % 8) return YV1_ERR_BAD_ARG;
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(k_avgpool_fwd, dim3(ew_blocks(total)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, N, H, W, C);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_avgpool2_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int C, hipStream_t stream) {
  if (!dy || !dx || N <= 0 || C % 8 || lddy % 8 || lddx % 8) return YV1_ERR_BAD_ARG;
  const long long total = (long long)N * H * W * (C / 8);
  hipLaunchKernelGGL(k_avgpool_bwd, dim3(ew_blocks(total)), dim3(256), 0, stream, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, N, H, W, C);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_head_sigmoid_fwd(const void* y, int ldy, const float* scale, const float* shift, float* out,
                                    long long npix, int C, hipStream_t stream) {
  if (!y || !scale || !shift || !out || npix <= 0 || C <= 0) return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_head_fwd, dim3(ew_blocks(npix * C)), dim3(256), 0, stream, (const bf16_t*)y, ldy, scale, shift, out, npix, C);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// dy gets lddy channels written (channels >= C zeroed)
extern "C" int yv1_head_sigmoid_bwd(const float* dout, const float* out, const void* y, int ldy, const float* gamma,
                                    const float* mean, const float* invstd, void* dy, int lddy, float* dgamma, float* dbeta,
                                    long long npix, int C, hipStream_t stream) {
  if (!dout || !out || !y || !gamma || !mean || !invstd || !dy || !dgamma || !dbeta || npix <= 0 || C <= 0 || lddy < C)
    return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_head_bwd, dim3(lddy), dim3(256), 0, stream, dout, out, (const bf16_t*)y, ldy, gamma, mean, invstd,
                     (bf16_t*)dy, lddy, dgamma, dbeta, npix, C);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_prep_weights(const float* w, long long so, long long si, long long sh, long long sw, int O, int I, int KH,
                                int KW, int Opad, int Ipad, void* dst_fwd, void* dst_t, hipStream_t stream) {
  if (!w || !dst_fwd || O <= 0 || I <= 0 || Opad < O || Ipad < I) return YV1_ERR_BAD_ARG;
  const long long total = (long long)Opad * KH * KW * Ipad;
  hipLaunchKernelGGL(k_prep_weights, dim3(ew_blocks(total)), dim3(256), 0, stream, w, so, si, sh, sw, O, I, KH, KW, Opad, Ipad,
                     (bf16_t*)dst_fwd, (bf16_t*)dst_t);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_prep_weights_max_tensors(void) { return PREP_MAX; }

// Same as yv1_prep_weights for `count` (<= yv1_prep_weights_max_tensors()) square-kernel weights in ONE launch.
// All arrays are HOST arrays; dst_t[i] may be NULL.
extern "C" int yv1_prep_weights_multi(const float* const* w, const long long* strides4, const int* O, const int* I, const int* K,
                                      const int* Opad, const int* Ipad, void* const* dst_fwd, void* const* dst_t, int count,
                                      hipStream_t stream) {
  if (!w || !strides4 || !O || !I || !K || !Opad || !Ipad || !dst_fwd || !dst_t || count <= 0 || count > PREP_MAX)
    return YV1_ERR_BAD_ARG;
  PrepTable t;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (!w[i] || !dst_fwd[i] || O[i] <= 0 || I[i] <= 0 || K[i] <= 0 || Opad[i] < O[i] || Ipad[i] < I[i]) return YV1_ERR_BAD_ARG;
    t.w[i] = w[i]; t.dst_fwd[i] = (bf16_t*)dst_fwd[i]; t.dst_t[i] = (bf16_t*)dst_t[i];
    t.so[i] = strides4[4 * i]; t.si[i] = strides4[4 * i + 1]; t.sh[i] = strides4[4 * i + 2]; t.sw[i] = strides4[4 * i + 3];
    t.O[i] = O[i]; t.I[i] = I[i]; t.K[i] = K[i]; t.Opad[i] = Opad[i]; t.Ipad[i] = Ipad[i];
    t.first_block[i] = blocks;
    blocks += ((Opad[i] + PREP_TILE - 1) / PREP_TILE) * ((Ipad[i] + PREP_TILE - 1) / PREP_TILE) * K[i] * K[i];
  }
  t.first_block[count] = blocks;
  t.count = count;
  hipLaunchKernelGGL(k_prep_weights_multi, dim3(blocks), dim3(256), 0, stream, t);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_prep_stem_weights(const float* w, long long so, long long si, long long sh, long long sw, int O, void* dst,
                                     hipStream_t stream) {
  if (!w || !dst || O <= 0) return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_prep_stem, dim3(ew_blocks((long long)O * 224)), dim3(256), 0, stream, w, so, si, sh, sw, O, (bf16_t*)dst);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_unpack_stem_grad(const float* g, float* dw, long long so, long long si, long long sh, long long sw, int O,
                                    hipStream_t stream) {
  if (!g || !dw || O <= 0) return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_unpack_stem_grad, dim3(ew_blocks((long long)O * 147)), dim3(256), 0, stream, g, dw, so, si, sh, sw, O);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}
