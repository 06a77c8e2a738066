// Implicit-GEMM convolution on MFMA for gfx950 (MI355X / CDNA4): forward and data-gradient.
//
// Replaces what the reference gets from cuDNN behind nn.Conv2d: conv3x3/conv1x1
// (backbones/OriginResNet.py:21-29), the 7x7/2 stem (:121), the projection shortcuts (:159-163),
// the DenseNet 1x1/3x3 layers (backbones/OriginDenseNet.py:24-29, :52-53) and their dgrad.
//
// GEMM view:  D[m][n] = sum_k A[m][k] * B[k][n]
//   m = GEMM pixel (img, p, q)            M = N*P*Q
//   n = output channel                    (weights stored [Cout][taps][Cin]: K contiguous)
//   k = (tap r,s ; input channel c)       K = R*S*Cin, walked tap-major in BK-wide channel blocks
// A is never materialised: each K-step gathers BM pixel rows x BK channels of ONE tap straight
// from the NHWC activation (16-B chunks = 8 bf16 channels), zero-filling padded taps.  A generic
// integer tap map   ih = (p*ah + r*bh + ch) / d  (valid iff divisible and in range)
// expresses forward (a=stride,b=1,c=-pad,d=1) and dgrad (a=1,b=-1,c=pad,d=stride, weights
// pre-transposed to [Cin][taps][Cout]) with the same kernel.  The 7x7/2 stem runs as R=7 taps of
// 32 contiguous elements over a zero-padded NHWC4 image (see yv1_pack_input_nhwc4).
//
// Tiling (wave64, v_mfma_f32_32x32x16_bf16): 256 threads = 4 wavefronts per workgroup, each wave
// owns a (BM/WM)x(BN/WN) sub-tile as 32x32 accumulator blocks; the LDS image of a tile has its 16-B
// chunks XOR-swizzled so every ds_read_b128 fragment read is bank-conflict-free; one barrier per K-step.
// Two main loops share everything else (tiles, swizzle, epilogue):
//   k_conv_dma  (default) tiles travel global -> LDS by LDS-DMA into a ring of 2-3 stages, the next K-steps stay in
//               flight across a raw s_barrier (counted s_waitcnt vmcnt) -- see the comment above the kernel;
//   k_conv_gemm register-staged (global_load_dwordx4 one K-step ahead, ds_write_b128 after the MFMA block),
//               double-buffered: the stem, 32-wide Cout tiles with Cin % 64 != 0, and YV1_CONV_DMA=0.
// Epilogue: per-channel sum / sum-of-squares of the fp32 accumulators (training-mode BatchNorm
// statistics, nn.BatchNorm2d at OriginResNet.py:123 etc.) reduced in-register + one shuffle and
// written as per-M-tile partials (deterministic, no atomics); the tile is packed to bf16 through
// LDS and stored as full 16-B channel runs (optionally accumulating into the destination, used by
// the strided 1x1 shortcut dgrad, adding the ReLU-masked shortcut gradient, or -- inference -- applying the
// folded eval-mode BatchNorm, the residual add and the ReLU).
// Workgroup -> tile map is XCD-aware: the workgroups that share an XCD (blockIdx % 8) walk the
// Cout tiles of the same pixel tile back to back, so the gathered A rows are re-read from that
// XCD's L2, not from HBM.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct ConvArgs {
  const bf16_t* X;
  const bf16_t* W;
  bf16_t* Y;
  float* stats = nullptr;   // [MT][2][Cout] partial sums, or nullptr
  int N, IH, IW, ldx;  // source tensor: pixel stride ldx elements
  int P, Q;            // GEMM pixel grid per image
  int Cin, Cout;       // K per tap, GEMM N
  int R, S;
  int ah, bh, ch, aw, bw, cw, log2d;
  int OH, OW, ldy, os; // destination tensor: GEMM pixel (n,p,q) -> (n, p*os + oh0, q*os + ow0)
  int oh0, ow0;
  int wr0, wrs, ws0, wss, WS;   // weight tap of loop tap (r,s): (wr0 + r*wrs)*WS + (ws0 + s*wss)
  int Kw;                       // elements per weight row (= all filter taps x Cin)
  int accumulate = 0;
  const bf16_t* AS = nullptr;   // optional: out += AM-bit ? AS[m][n] : 0 (shortcut gradient through a ReLU, see
  const unsigned char* AM = nullptr;   //        yv1_conv2d_dgrad_add_masked_nhwc_bf16); AM is [M][ldam] bytes, bit k of
  int ldas = 0, ldam = 0;       //           byte j = channel 8j+k
  const float* escale = nullptr;   // optional inference epilogue: t = bf16(acc * escale[n] + eshift[n]) (folded eval-mode
  const float* eshift = nullptr;   //   BatchNorm), out = t + ERES (bf16 residual), ReLU if erelu
  const bf16_t* ERES = nullptr;
  int ldres = 0, erelu = 0;
  // round 3, "bn3 as algebra" (DESIGN.md section 7):
  const bf16_t* X2 = nullptr;   // 1x1 direct launches: K-steps >= cb_split read their channels from this second source
  int ldx2 = 0, cb_split = 0;   //   tensor (pixel stride ldx2) -- the K-concatenation [X | X2] without a copy
  const unsigned char* OM = nullptr;   // optional OUTPUT mask [dest pixels][ldom bytes] (bit k of byte j = channel 8j+k): channels
  int ldom = 0;                        //   whose bit is 0 are stored as zero (the ReLU mask of the block whose output gradient this is)
  float* gsum = nullptr;               // optional [MT][Cout] per-tile column sums of the values this launch ADDED to the destination
  // round 3, "deferred BatchNorm backward" (DenseNet norm1 / transition norm): the launch's output is the
  // gradient at the OUTPUT of relu(bn(x)); the epilogue masks it (scale*x + shift > 0), adds scale * masked to the destination
  // (the reduction-free term of the BatchNorm backward) and emits the per-tile sums the finalize needs
  const bf16_t* DBX = nullptr;         // the BatchNorm's INPUT x [M][lddbx] (k_conv_dma<..., DB = true> only)
  int lddbx = 0;
  const float* dbscale = nullptr;      // forward scale / shift / mean of that BatchNorm per GEMM column
  const float* dbshift = nullptr;
  const float* dbmean = nullptr;
  float* dbpart = nullptr;             // [MT][2][Cout]: sum(d), sum(d * (x - mean)) with d = masked gradient (bf16 values)
  int db_wt_rows = 0;                  // readable rows of W (zero beyond Cout): lets the last column tile reach past Cout
  const float* dbpa = nullptr;         // optional PENDING correction of the destination (deferred form): dx -= dbpa[c] + dbpb[c] * x,
  const float* dbpb = nullptr;         //   the affine term the BatchNorm handled by the PREVIOUS launch into this buffer still owes
  int db_unit = 0;                     // 1: store the masked gradient itself (not scale * masked): the BatchNorm backward's
  const float* dbis = nullptr;         //    apply pass follows; dbis: invstd -- the second sum is then sum(d * xhat)
  int M;
  int MT, NT;
  int slots = 0;       // k_conv_ps: workgroups per column tile; workgroup (slot, nt) walks the pixel tiles slot, slot+slots, ...
  int stagger = 0, ncu = 256;   // tuning: workgroup b starts (b / ncu) * stagger * ~0.4 us late (co-resident workgroups out of phase)
  int dbg = 0;         // tuning only: bit0 skip the in-loop global loads / LDS stores, bit1 skip the MFMA block
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <int BK>
__device__ __forceinline__ int swz(int row, int chunk) {
  constexpr int CPR = BK / 8;                 // 16-B chunks per row
  return chunk ^ ((row / (16 / CPR)) % CPR);  // conflict-free for the 32x32x16 fragment reads
}

// Shared epilogue of the GEMM kernels: BatchNorm statistic partials from the fp32 accumulators, DPP lane swap +
// packed rounding into a bf16 LDS tile, full-line stores (optionally accumulating / adding the masked shortcut
// gradient).  The caller has finished its last LDS read (barrier) before the tile overwrites the staging buffers.
template <int BM, int BN, int WM, int WN, bool DB = false, bool PRE = false>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const ConvArgs& a,
                                              unsigned char* smem, int m0, int n0, int mt) {
  constexpr int NTH = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int EPI_PITCH = (BN / 2 % 32 == 16) ? BN * 2 : BN * 2 + 64;   // bytes; pitch/4 % 32 == 16
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  // ---- DB epilogue: its operands (the BatchNorm input x and the destination's old values) are requested HERE, before the
  // accumulators go through LDS -- their latency runs under epilogue 2 and its barrier instead of after it
  constexpr int DB_OCPR = BN / 8, DB_PASSES = (BM * DB_OCPR + NTH - 1) / NTH, DB_RG = NTH / DB_OCPR;
  uint4 x_pre[DB ? DB_PASSES : 1], old_pre[DB ? DB_PASSES : 1];
  if constexpr (DB) {
    const int cc = tid % DB_OCPR, rg = tid / DB_OCPR;
    const bool cok = n0 + cc * 8 < a.Cout;
#pragma unroll
    for (int i = 0; i < DB_PASSES; ++i) {
      const int row = rg + i * DB_RG, m = m0 + row;
      x_pre[i] = make_uint4(0u, 0u, 0u, 0u); old_pre[i] = make_uint4(0u, 0u, 0u, 0u);
      if (cok && row < BM && m < a.M) {
        x_pre[i] = *reinterpret_cast<const uint4*>(a.DBX + (size_t)m * a.lddbx + n0 + cc * 8);
        if (a.accumulate) old_pre[i] = *reinterpret_cast<const uint4*>(a.Y + (size_t)m * a.ldy + n0 + cc * 8);
      }
    }
  }
  // ---- generic epilogue operands (shortcut gradient + masks, or the destination's old values): with PRE (k_conv_dma: one tile
  // per workgroup, nothing else to overlap with) they are requested before the accumulators go through LDS, like the DB ones
  constexpr int G_OCPR = BN / 8, G_PASSES = (BM * G_OCPR + NTH - 1) / NTH;
  uint4 as_pre[G_PASSES];
  unsigned am_pre[G_PASSES], om_pre[G_PASSES];
  auto prefetch_generic = [&]() {
    if (a.OM) {
  #pragma unroll
      for (int i = 0; i < G_PASSES; ++i) {
        const int idx = tid + i * NTH;
        const int row = idx / G_OCPR, cc = idx - row * G_OCPR;
        const int m = m0 + row;
        om_pre[i] = 0xffu;
        if (row < BM && m < a.M) {
          size_t dp = (size_t)m;
          if (a.os != 1) {
            const int pq = a.P * a.Q;
            const int n = m / pq, rem = m - n * pq;
            const int p = rem / a.Q, q = rem - p * a.Q;
            dp = (size_t)(n * a.OH + p * a.os + a.oh0) * a.OW + q * a.os + a.ow0;
          }
          om_pre[i] = a.OM[dp * a.ldom + ((n0 + cc * 8) >> 3)];
        }
      }
    }
    if (a.AS) {
  #pragma unroll
      for (int i = 0; i < G_PASSES; ++i) {
        const int idx = tid + i * NTH;
        const int row = idx / G_OCPR, cc = idx - row * G_OCPR;
        const int m = m0 + row;
        if (row < BM && m < a.M) {
          as_pre[i] = *reinterpret_cast<const uint4*>(a.AS + (size_t)m * a.ldas + n0 + cc * 8);
          am_pre[i] = a.AM ? a.AM[(size_t)m * a.ldam + ((n0 + cc * 8) >> 3)] : 0xffu;     // no mask: AS is added as it is
        }
      }
    } else if (a.accumulate) {                 // the destination's old values, likewise (AS and accumulate are never combined)
  #pragma unroll
      for (int i = 0; i < G_PASSES; ++i) {
        const int idx = tid + i * NTH;
        const int row = idx / G_OCPR, cc = idx - row * G_OCPR;
        const int m = m0 + row;
        if (row < BM && m < a.M) {
          size_t off;
          if (a.os == 1) {
            off = (size_t)m * a.ldy + n0 + cc * 8;
          } else {
            const int pq = a.P * a.Q;
            const int n = m / pq, rem = m - n * pq;
            const int p = rem / a.Q, q = rem - p * a.Q;
            off = ((size_t)(n * a.OH + p * a.os + a.oh0) * a.OW + q * a.os + a.ow0) * a.ldy + n0 + cc * 8;
          }
          as_pre[i] = *reinterpret_cast<const uint4*>(a.Y + off);
        }
      }
    }
  };
  const bool pre_now = PRE && !DB && !(a.dbg & 4);          // YV1_CONV_DBG bit 2 (tuning): request them after the barrier, as before
  if (pre_now) prefetch_generic();
  // ---- epilogue 1: BatchNorm batch statistics from the fp32 accumulators
  // C/D layout of 32x32: col = lane&31 (channel), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel)
  float* red = reinterpret_cast<float*>(smem + BM * EPI_PITCH);   // [WM][2][BN] floats, after the epilogue tile
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc[i][j][e];
          s += v;
          ss += v * v;
        }
      s += __shfl_xor(s, 32, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (lh == 0) {
        const int c = wn * (BN / WN) + j * 32 + l31;
        red[(wm * 2 + 0) * BN + c] = s;
        red[(wm * 2 + 1) * BN + c] = ss;
      }
    }
  }

  // ---- epilogue 2: accumulators -> bf16 tile in LDS.  A lane holds ONE channel (column) of 16 pixel rows;
  // neighbouring lanes swap one value (DPP quad_perm, no LDS traffic) so each lane owns a 2-channel pair,
  // rounded by v_cvt_pk_bf16_f32 and written as one dword.
  unsigned char* et = smem;
  const bool odd = lane & 1;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wn * (BN / WN) + j * 32 + l31;
      const int rbase = wm * (BM / WM) + i * 32 + 4 * lh;
      const bool affine = a.escale != nullptr || a.erelu;   // training launches (raw conv output) skip scale/shift/clamp
      const float al = a.escale ? a.escale[n0 + col] : 1.f, be = a.escale ? a.eshift[n0 + col] : 0.f;
      const float lo_clamp = (a.erelu && !a.ERES) ? 0.f : -3.0e38f;     // ReLU here unless a residual is added first
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const float mine_lo = affine ? fmaxf(acc[i][j][e] * al + be, lo_clamp) : acc[i][j][e];
        const float mine_hi = affine ? fmaxf(acc[i][j][e + 1] * al + be, lo_clamp) : acc[i][j][e + 1];
        const float send = odd ? mine_lo : mine_hi;
        const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xf, 0xf, true));
        // even lane: row(e), channels (col, col+1) = (mine_lo, neighbour's acc[e]); odd: row(e+1), (col-1, col)
        const int row = rbase + ((e + (odd ? 1 : 0)) & 3) + 8 * (e >> 2);
        const unsigned v = odd ? cvt_pk_bf16(recv, mine_hi) : cvt_pk_bf16(mine_lo, recv);
        *reinterpret_cast<unsigned*>(et + row * EPI_PITCH + (col & ~1) * 2) = v;
      }
      __builtin_amdgcn_sched_barrier(0);   // one 32x32 block at a time: keeps the accumulator->VGPR copies short-lived
    }
  // DB epilogue: the tile's per-channel vectors (scale, shift, mean, pending A / B) go through LDS once per workgroup, behind
  // the statistics area -- fetched per thread they were 160 B x 256 threads = 40 KB of L2 requests per tile, as much as the
  // tile's own operands (the deferred data gradient ran 298 us against a 177 us byte bound at 112x112, 13-25 % more with the
  // pending vectors)
  float* dbv = reinterpret_cast<float*>(smem + BM * EPI_PITCH + WM * 2 * BN * 4);     // [5][BN]
  if constexpr (DB) {
    if (tid < BN) {
      const int c = n0 + tid;
      const bool ok = c < a.Cout;
      dbv[0 * BN + tid] = ok ? a.dbscale[c] : 0.f;
      dbv[1 * BN + tid] = ok ? a.dbshift[c] : 0.f;
      dbv[2 * BN + tid] = ok ? a.dbmean[c] : 0.f;
      dbv[3 * BN + tid] = (ok && a.dbpa) ? a.dbpa[c] : 0.f;
      dbv[4 * BN + tid] = (ok && a.dbpa) ? a.dbpb[c] : 0.f;
    }
  }
  __syncthreads();

  if (a.stats && tid < BN) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) {
      s += red[(w * 2 + 0) * BN + tid];
      ss += red[(w * 2 + 1) * BN + tid];
    }
    float* o = a.stats + (size_t)mt * 2 * a.Cout + n0 + tid;
    o[0] = s;
    o[a.Cout] = ss;
  }

  // ---- epilogue 3: full-line stores, 16 B (8 channels) per lane
  constexpr int OCPR = BN / 8;
  constexpr int OPASSES = (BM * OCPR + NTH - 1) / NTH;
  if constexpr (DB) {
    // deferred BatchNorm backward (see ConvArgs::DBX): destination pixels are the GEMM pixels (os == 1)
    static_assert(NTH % OCPR == 0, "a thread keeps its channel chunk across the store passes");
    constexpr int RG = NTH / OCPR;
    const int cc = tid % OCPR, rg = tid / OCPR;
    const bool cok = n0 + cc * 8 < a.Cout;                 // the last column tile may reach past Cout (zero weight rows)
    // scale / shift stay in registers (every element needs them twice); the pending vectors are re-read from LDS per pass
    // and the mean enters once, at the end (sum d*(x - mu) = sum d*x - mu * sum d over this thread's <= 8 rows): 24 VGPRs
    // less, which is what lets the 128x64 form run five workgroups per CU
    float sc[8], sh[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sc[k] = dbv[0 * BN + cc * 8 + k]; sh[k] = dbv[1 * BN + cc * 8 + k];
      s1[k] = 0.f; s2[k] = 0.f;
    }
    const bool pend = a.dbpa != nullptr;
    // 128-wide tiles (eight store passes, 168 VGPRs allowed): the pending vectors in registers after all -- re-read per pass
    // they cost that form 10 % (314 -> 344 us at 112x112, Cin 256)
    constexpr bool PREG = BN >= 128;
    float pa[PREG ? 8 : 1], pb[PREG ? 8 : 1];
    if constexpr (PREG) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { pa[k] = dbv[3 * BN + cc * 8 + k]; pb[k] = dbv[4 * BN + cc * 8 + k]; }
    }
    static_assert(OPASSES == DB_PASSES && RG == DB_RG, "the hoisted operand loads use the same pass geometry");
#pragma unroll
    for (int i = 0; i < OPASSES; ++i) {
      const int row = rg + i * RG, m = m0 + row;
      if (cok && row < BM && m < a.M) {
        const uint4 v = *reinterpret_cast<const uint4*>(et + row * EPI_PITCH + cc * 16);
        const unsigned* pv = reinterpret_cast<const unsigned*>(&v);
        const unsigned* px = reinterpret_cast<const unsigned*>(&x_pre[i]);
        const unsigned* po = reinterpret_cast<const unsigned*>(&old_pre[i]);
        unsigned res[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xl = __uint_as_float(px[k] << 16), xh = __uint_as_float(px[k] & 0xffff0000u);
          const float dl = (xl * sc[2 * k] + sh[2 * k]) > 0.f ? __uint_as_float(pv[k] << 16) : 0.f;
          const float dh = (xh * sc[2 * k + 1] + sh[2 * k + 1]) > 0.f ? __uint_as_float(pv[k] & 0xffff0000u) : 0.f;
          s1[2 * k] += dl; s1[2 * k + 1] += dh;
          s2[2 * k] += dl * xl; s2[2 * k + 1] += dh * xh;
          float ol = __uint_as_float(po[k] << 16) + (a.db_unit ? dl : sc[2 * k] * dl);
          float oh = __uint_as_float(po[k] & 0xffff0000u) + (a.db_unit ? dh : sc[2 * k + 1] * dh);
          if constexpr (PREG) {
            ol -= pa[2 * k] + pb[2 * k] * xl;
            oh -= pa[2 * k + 1] + pb[2 * k + 1] * xh;
          } else if (pend) {
            ol -= dbv[3 * BN + cc * 8 + 2 * k] + dbv[4 * BN + cc * 8 + 2 * k] * xl;
            oh -= dbv[3 * BN + cc * 8 + 2 * k + 1] + dbv[4 * BN + cc * 8 + 2 * k + 1] * xh;
          }
          res[k] = pack_bf16x2(ol, oh);
        }
        *reinterpret_cast<uint4*>(a.Y + (size_t)m * a.ldy + n0 + cc * 8) = make_uint4(res[0], res[1], res[2], res[3]);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s2[k] -= dbv[2 * BN + cc * 8 + k] * s1[k];       // sum d*(x - mu) of this thread's rows
    // fixed-order reduction over the RG threads that hold the same channel chunk; the epilogue tile is dead
    __syncthreads();
    float* r1 = reinterpret_cast<float*>(smem);            // [RG][BN]
    float* r2 = r1 + RG * BN;
#pragma unroll
    for (int k = 0; k < 8; ++k) { r1[rg * BN + cc * 8 + k] = s1[k]; r2[rg * BN + cc * 8 + k] = s2[k]; }
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
      for (int r = 0; r < RG; ++r) { t1 += r1[r * BN + tid]; t2 += r2[r * BN + tid]; }
      float* o = a.dbpart + (size_t)mt * 2 * a.Cout + n0 + tid;
      o[0] = t1;
      o[a.Cout] = a.dbis ? t2 * a.dbis[n0 + tid] : t2;
    }
    return;
  }
  // shortcut-gradient operands of ALL passes up front: inside the loop every load sits behind the previous pass's store
  // (the compiler cannot prove Y and AS apart), i.e. one memory round trip per pass instead of one per tile
  float gs[8];                                              // this thread's column sums (it keeps one 8-channel chunk: NTH % OCPR == 0)
#pragma unroll
  for (int k = 0; k < 8; ++k) gs[k] = 0.f;
  if (!pre_now) prefetch_generic();
#pragma unroll
  for (int i = 0; i < OPASSES; ++i) {
    const int idx = tid + i * NTH;
    const int row = idx / OCPR, cc = idx - row * OCPR;
    const int m = m0 + row;
    if (row < BM && m < a.M) {
      size_t off;
      if (a.os == 1) {                       // destination pixels are the GEMM pixels in order
        off = (size_t)m * a.ldy + n0 + cc * 8;
      } else {
        const int pq = a.P * a.Q;
        const int n = m / pq, rem = m - n * pq;
        const int p = rem / a.Q, q = rem - p * a.Q;
        off = ((size_t)(n * a.OH + p * a.os + a.oh0) * a.OW + q * a.os + a.ow0) * a.ldy + n0 + cc * 8;
      }
      uint4 v = *reinterpret_cast<const uint4*>(et + row * EPI_PITCH + cc * 16);
      if (a.AS) {
        const uint4 o = as_pre[i];
        const unsigned mb = am_pre[i];
        const unsigned* pv = reinterpret_cast<const unsigned*>(&v);
        const unsigned* po = reinterpret_cast<const unsigned*>(&o);
        unsigned res[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float lo = __uint_as_float(pv[k] << 16) + (((mb >> (2 * k)) & 1u) ? __uint_as_float(po[k] << 16) : 0.f);
          const float hi = __uint_as_float(pv[k] & 0xffff0000u) +
                           (((mb >> (2 * k + 1)) & 1u) ? __uint_as_float(po[k] & 0xffff0000u) : 0.f);
          res[k] = pack_bf16x2(lo, hi);
        }
        v = make_uint4(res[0], res[1], res[2], res[3]);
      }
      if (a.ERES) {
        const uint4 o = *reinterpret_cast<const uint4*>(a.ERES + (size_t)m * a.ldres + n0 + cc * 8);
        const unsigned* pv = reinterpret_cast<const unsigned*>(&v);
        const unsigned* po = reinterpret_cast<const unsigned*>(&o);
        const float lo_clamp = a.erelu ? 0.f : -3.0e38f;
        unsigned res[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float lo = fmaxf(__uint_as_float(pv[k] << 16) + __uint_as_float(po[k] << 16), lo_clamp);
          const float hi = fmaxf(__uint_as_float(pv[k] & 0xffff0000u) + __uint_as_float(po[k] & 0xffff0000u), lo_clamp);
          res[k] = pack_bf16x2(lo, hi);
        }
        v = make_uint4(res[0], res[1], res[2], res[3]);
      }
      uint4 oldv = make_uint4(0u, 0u, 0u, 0u);
      if (a.accumulate) {
        const uint4 o = a.AS ? *reinterpret_cast<const uint4*>(a.Y + off) : as_pre[i];
        oldv = o;
        const unsigned* pv = reinterpret_cast<const unsigned*>(&v);
        const unsigned* po = reinterpret_cast<const unsigned*>(&o);
        unsigned res[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float lo = __uint_as_float(pv[k] << 16) + __uint_as_float(po[k] << 16);
          const float hi = __uint_as_float(pv[k] & 0xffff0000u) + __uint_as_float(po[k] & 0xffff0000u);
          res[k] = pack_bf16x2(lo, hi);
        }
        v = make_uint4(res[0], res[1], res[2], res[3]);
      }
      if (a.OM) {                                         // zero the channels the destination block's ReLU closed
        const unsigned mb = om_pre[i];
        unsigned* pv = reinterpret_cast<unsigned*>(&v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned keep = (((mb >> (2 * k)) & 1u) ? 0x0000ffffu : 0u) | (((mb >> (2 * k + 1)) & 1u) ? 0xffff0000u : 0u);
          pv[k] &= keep;
        }
      }
      if (a.gsum) {                                       // column sums of what this launch added: stored - old (bf16 values)
        const unsigned* pv = reinterpret_cast<const unsigned*>(&v);
        const unsigned* po = reinterpret_cast<const unsigned*>(&oldv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          gs[2 * k] += __uint_as_float(pv[k] << 16) - __uint_as_float(po[k] << 16);
          gs[2 * k + 1] += __uint_as_float(pv[k] & 0xffff0000u) - __uint_as_float(po[k] & 0xffff0000u);
        }
      }
      *reinterpret_cast<uint4*>(a.Y + off) = v;
    }
  }
  if (a.gsum) {
    // fixed-order reduction over the NTH / OCPR threads that hold the same channel chunk; the epilogue tile is dead
    static_assert(NTH % OCPR == 0, "a thread keeps its channel chunk across the store passes");
    constexpr int RG = NTH / OCPR;
    __syncthreads();
    float* gred = reinterpret_cast<float*>(smem);          // [RG][BN]
    const int cc = tid % OCPR, rg = tid / OCPR;
#pragma unroll
    for (int k = 0; k < 8; ++k) gred[rg * BN + cc * 8 + k] = gs[k];
    __syncthreads();
    if (tid < BN) {
      float t = 0.f;
#pragma unroll 8
      for (int r = 0; r < RG; ++r) t += gred[r * BN + tid];
      a.gsum[(size_t)mt * a.Cout + n0 + tid] = t;
    }
  }
}

template <int BM, int BN, int BK, int WM, int WN>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN == 4 ? 3 : 1)) k_conv_gemm(ConvArgs a) {
  constexpr int NTH = WM * WN * 64;           // threads per workgroup (one wave per (wm, wn))
  constexpr int CPR = BK / 8;
  constexpr int ROWS_PER_PASS = NTH / CPR;
  constexpr int A_PASSES = (BM + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
  constexpr int B_PASSES = (BN + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
  constexpr bool A_GUARD = (BM % ROWS_PER_PASS) != 0, B_GUARD = (BN % ROWS_PER_PASS) != 0;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;

  // ---- XCD-aware tile assignment (bijective; blocks b and b+8 share an XCD)
  int mt, nt;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    mt = lin / a.NT;
    nt = lin - mt * a.NT;
  }
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread loader state
  const int ccol = tid % CPR;          // chunk column inside the K-step
  const int rrow = tid / CPR;          // first row handled
  int pix_base[A_PASSES];              // image base (n*IH*IW) or -1 when the row is past M / past the tile
  int ph[A_PASSES], qw[A_PASSES];      // p*ah + ch , q*aw + cw
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + rrow + i * ROWS_PER_PASS;
    if (m < a.M && (!A_GUARD || rrow + i * ROWS_PER_PASS < BM)) {
      const int pq = a.P * a.Q;
      const int n = m / pq, rem = m - n * pq;
      const int p = rem / a.Q, q = rem - p * a.Q;
      pix_base[i] = n * a.IH * a.IW;
      ph[i] = p * a.ah + a.ch;
      qw[i] = q * a.aw + a.cw;
    } else {
      pix_base[i] = -1; ph[i] = 0; qw[i] = 0;
    }
  }
  const int Ktot = a.Kw;                  // a weight row always holds the full filter, whatever taps this launch walks
  const int cblocks = a.Cin / BK;
  const int nk = a.R * a.S * cblocks;
  const int dmask = (1 << a.log2d) - 1;
  // weights: row base per thread; the tap / channel-block offset is added per load
  // (32-bit element offsets: every tensor here is far below 2^31 elements; keeps the VGPR count down)
  int woff[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int row = rrow + i * ROWS_PER_PASS;
    woff[i] = (n0 + ((!B_GUARD || row < BN) ? row : 0)) * Ktot + ccol * 8;
  }

  // The loader runs one K-step ahead of the MFMA block.  Its tap state (source offset of each row
  // for the current filter tap, or "padding") is recomputed only when the tap changes, not per step.
  int ld_r = 0, ld_s = 0, ld_cb = 0;
  int wtap_off = 0;                    // element offset of the current tap inside a weight row
  int aoff[A_PASSES];                  // element offset of the tapped pixel + chunk column; 0 when the tap is padding
  unsigned avalid = 0;                 // bit i: row i of this thread reads real data for the current tap
#define YV1_SET_TAP()                                                                                            \
  {                                                                                                              \
    avalid = 0;                                                                                                  \
    wtap_off = ((a.wr0 + ld_r * a.wrs) * a.WS + (a.ws0 + ld_s * a.wss)) * a.Cin;                                 \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
      const int hn = ph[i] + ld_r * a.bh, wn_ = qw[i] + ld_s * a.bw;                                             \
      const int ih = hn >> a.log2d, iw = wn_ >> a.log2d;                                                         \
      const bool ok = pix_base[i] >= 0 && ((hn | wn_) & dmask) == 0 && hn >= 0 && wn_ >= 0 && ih < a.IH &&       \
                      iw < a.IW;                                                                                 \
      aoff[i] = ok ? (pix_base[i] + ih * a.IW + iw) * a.ldx + ccol * 8 : 0;                                      \
      avalid |= ok ? (1u << i) : 0u;                                                                             \
    }                                                                                                            \
  }
  u32x4 ra[A_PASSES], rb[B_PASSES];
  // branch-free: padded taps load a dummy (valid) address and are zeroed with a select
#define YV1_LOAD_TILES()                                                                                         \
  {                                                                                                              \
    const int coff = ld_cb * BK;                                                                                 \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
      const bool ok = (avalid >> i) & 1u;                                                                        \
      u32x4 v = *reinterpret_cast<const u32x4*>(a.X + (aoff[i] + (ok ? coff : 0)));                              \
      const u32x4 z = {0u, 0u, 0u, 0u};                                                                          \
      ra[i] = ok ? v : z;                                                                                        \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i) {                                                       \
      rb[i] = *reinterpret_cast<const u32x4*>(a.W + (woff[i] + wtap_off + coff));                                \
    }                                                                                                            \
    if (++ld_cb == cblocks) {                                                                                    \
      ld_cb = 0;                                                                                                 \
      if (++ld_s == a.S) { ld_s = 0; ++ld_r; }                                                                   \
      YV1_SET_TAP();                                                                                             \
    }                                                                                                            \
  }
#define YV1_STORE_TILES(BUF_)                                                                                    \
  {                                                                                                              \
    unsigned char* sa_ = smem + (BUF_) * STAGE;                                                                  \
    unsigned char* sb_ = sa_ + A_BYTES;                                                                          \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
      const int row = rrow + i * ROWS_PER_PASS;                                                                  \
      if (!A_GUARD || row < BM) *reinterpret_cast<u32x4*>(sa_ + row * (BK * 2) + swz<BK>(row, ccol) * 16) = ra[i]; \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i) {                                                       \
      const int row = rrow + i * ROWS_PER_PASS;                                                                  \
      if (!B_GUARD || row < BN) *reinterpret_cast<u32x4*>(sb_ + row * (BK * 2) + swz<BK>(row, ccol) * 16) = rb[i]; \
    }                                                                                                            \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  YV1_SET_TAP();
  YV1_LOAD_TILES();
  YV1_STORE_TILES(0);
  __syncthreads();

  const int l31 = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool do_ld = (kt + 1 < nk) && !(a.dbg & 1);
    if (do_ld) YV1_LOAD_TILES();                  // global loads in flight during the MFMA block
    const unsigned char* sa = smem + cur * STAGE;
    const unsigned char* sb = sa + A_BYTES;
    if (!(a.dbg & 2))
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / WM) + i * 32 + l31;
        fa[i] = *reinterpret_cast<const bf16x8*>(sa + row * (BK * 2) + swz<BK>(row, ks * 2 + lh) * 16);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (BN / WN) + j * 32 + l31;
        fb[j] = *reinterpret_cast<const bf16x8*>(sb + row * (BK * 2) + swz<BK>(row, ks * 2 + lh) * 16);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (do_ld) YV1_STORE_TILES(cur ^ 1);
    __syncthreads();
  }

  conv_epilogue<BM, BN, WM, WN>(acc, a, smem, m0, n0, mt);
}

// ---- LDS-DMA variant of the main loop -----------------------------------------------------------------------
// Same tiles, same LDS image, same fragment reads and epilogue as k_conv_gemm, but the A/B tiles travel
// global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write) into a ring of NST stages, with the
// loads of the next NST-1 K-steps in flight ACROSS the per-step barrier: a counted s_waitcnt vmcnt retires only the
// stage about to be read, and the barrier is a raw s_barrier (a __syncthreads() would drain the whole queue).
// The register-staged loop exposes one global-load latency per K-step and per workgroup (load -> MFMA -> wait ->
// ds_write -> barrier); here a workgroup's K-step costs its MFMAs plus one barrier.
// An LDS-DMA instruction writes wave-uniform base + lane*16 B, i.e. 1 KB of consecutive tile rows: the XOR swizzle is
// applied to the SOURCE chunk each lane fetches (slot s of row r holds logical chunk s ^ key(r)), the fragment reads
// use the same involution.  Padding taps fetch from a zero page (the DMA cannot select a constant).
__device__ __attribute__((aligned(16))) unsigned g_zero_page[4];

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  // gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4; expcnt imm[6:4] and lgkmcnt imm[11:8] left at "no wait"
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

template <int BM, int BN, int BK, int WM, int WN, int NST, bool DB = false>
__global__ void __launch_bounds__(WM * WN * 64, (BM == 256 ? 1 : (DB && BM == 128 && BN == 64 && NST == 2 ? 5 : (BM * BN <= 128 * 128 ? 3 : 2))))
k_conv_dma(ConvArgs a) {
  constexpr int NTH = WM * WN * 64;
  constexpr int CPR = BK / 8;
  constexpr int RPP = NTH / CPR;                       // rows per pass (one pass = one DMA instruction per wave)
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tiles must be whole passes");
  constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP, LPS = A_PASSES + B_PASSES;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA destinations stay in SGPRs
  const int wm = wid / WN, wn = wid % WN;
  int mt, nt;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    mt = lin / a.NT;
    nt = lin - mt * a.NT;
  }
  const int m0 = mt * BM, n0 = nt * BN;

  const int slot = tid % CPR, rrow = tid / CPR;
  const int lchunk = swz<BK>(rrow, slot);              // logical chunk this lane fetches (same key for every pass)
  int pix_base[A_PASSES], ph[A_PASSES], qw[A_PASSES];
  // pointwise, unit stride, no padding (most launches): GEMM row m IS input pixel m -- no (n, p, q) decomposition (two
  // integer divisions per pass); the row is addressed as pixel (0, m) of a one-row image that SET_TAP never clips
  const bool direct = a.R * a.S == 1 && a.ah == 1 && a.aw == 1 && a.ch == 0 && a.cw == 0 && a.log2d == 0 &&
                      a.P == a.IH && a.Q == a.IW;
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + rrow + i * RPP;
    if (m < a.M) {
      if (direct) {
        pix_base[i] = m; ph[i] = 0; qw[i] = 0;
      } else {
        const int pq = a.P * a.Q;
        const int n = m / pq, rem = m - n * pq;
        const int p = rem / a.Q, q = rem - p * a.Q;
        pix_base[i] = n * a.IH * a.IW;
        ph[i] = p * a.ah + a.ch;
        qw[i] = q * a.aw + a.cw;
      }
    } else {
      pix_base[i] = -1; ph[i] = 0; qw[i] = 0;
    }
  }
  const int cblocks = a.Cin / BK;
  const int nk = a.R * a.S * cblocks;
  const int dmask = (1 << a.log2d) - 1;
  int woff[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) woff[i] = (n0 + rrow + i * RPP) * a.Kw + lchunk * 8;
  // wave-uniform LDS byte offsets of this wave's 1 KB pieces inside a stage
  const int piece_row0 = (wid * 64) / CPR;

  // Loader state: one source pointer per DMA pass, advanced by BK channels per K-step inside a tap and recomputed when
  // the tap changes; a padding row points at the zero page and does not advance.  (The first version rebuilt every
  // address from integer offsets and re-selected the zero page in every K-step: ~130 scalar/vector instructions
  // around 8 MFMAs.)
  int ld_r = 0, ld_s = 0, ld_cb = 0;
  const bf16_t* asrc[A_PASSES];
  int astep[A_PASSES];
  const bf16_t* wsrc[B_PASSES];
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_zero_page);
#define YV1_SET_TAP_D()                                                                                          \
  {                                                                                                              \
    const int wtap_off = ((a.wr0 + ld_r * a.wrs) * a.WS + (a.ws0 + ld_s * a.wss)) * a.Cin;                       \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
      const int hn = ph[i] + ld_r * a.bh, wn_ = qw[i] + ld_s * a.bw;                                             \
      const int ih = hn >> a.log2d, iw = wn_ >> a.log2d;                                                         \
      const bool ok = pix_base[i] >= 0 && ((hn | wn_) & dmask) == 0 && hn >= 0 && wn_ >= 0 && ih < a.IH &&       \
                      iw < a.IW;                                                                                 \
      asrc[i] = ok ? a.X + ((size_t)(pix_base[i] + ih * a.IW + iw) * a.ldx + lchunk * 8) : zsrc;                 \
      astep[i] = ok ? BK : 0;                                                                                    \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i) wsrc[i] = a.W + (woff[i] + wtap_off);                   \
  }
#define YV1_ISSUE(STG_)                                                                                          \
  {                                                                                                              \
    unsigned char* sa_ = smem + (STG_) * STAGE;                                                                  \
    unsigned char* sb_ = sa_ + A_BYTES;                                                                          \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
      __builtin_amdgcn_global_load_lds((glb_void*)asrc[i], (lds_void*)(sa_ + (piece_row0 + i * RPP) * (BK * 2)), 16, 0, 0); \
      asrc[i] += astep[i];                                                                                       \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i) {                                                       \
      __builtin_amdgcn_global_load_lds((glb_void*)wsrc[i], (lds_void*)(sb_ + (piece_row0 + i * RPP) * (BK * 2)), 16, 0, 0); \
      wsrc[i] += BK;                                                                                             \
    }                                                                                                            \
    ++ld_cb;                                                                                                     \
    if (a.X2 && ld_cb == a.cb_split) {                   /* K-concatenation [X | X2]: direct 1x1 launches only */ \
      _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i)                                                       \
        asrc[i] = pix_base[i] >= 0 ? a.X2 + ((size_t)pix_base[i] * a.ldx2 + lchunk * 8) : zsrc;                  \
    }                                                                                                            \
    if (ld_cb == cblocks) {                                                                                      \
      ld_cb = 0;                                                                                                 \
      if (++ld_s == a.S) { ld_s = 0; ++ld_r; }                                                                   \
      YV1_SET_TAP_D();                                                                                           \
    }                                                                                                            \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  YV1_SET_TAP_D();
  // prologue: K-steps 0 .. NST-2 in flight
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < nk) YV1_ISSUE(p);

  // per-lane byte offsets of the MFMA fragments inside a stage (row and swizzled chunk): computed once, the stage base
  // is a compile-time constant in the unrolled loop below, so every fragment read is `ds_read_b128 v, offset:imm`
  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int KS = BK / 16;
  int fa_off[TM][KS], fb_off[TN][KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * (BM / WM) + i * 32 + l31;
      fa_off[i][ks] = row * (BK * 2) + swz<BK>(row, ks * 2 + lh) * 16;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * (BN / WN) + j * 32 + l31;
      fb_off[j][ks] = A_BYTES + row * (BK * 2) + swz<BK>(row, ks * 2 + lh) * 16;
    }
  }
#define YV1_MFMA_BLOCK(BASE_)                                                                                    \
  _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                            \
    bf16x8 fa[TM], fb[TN];                                                                                       \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>((BASE_) + fa_off[i][ks]); \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>((BASE_) + fb_off[j][ks]); \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                               \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                             \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);                   \
  }

  int kt = 0;
  // steady state, unrolled over the ring: stage indices are constants, NST-2 younger K-steps stay in flight, every
  // step issues the one NST-1 ahead.  Covers kt < nk - (NST-1) in groups of NST.
  if (!a.dbg) {
    const int n_main = nk - (NST - 1);
    for (; kt + NST <= n_main; kt += NST) {
#pragma unroll
      for (int c = 0; c < NST; ++c) {
        // every fragment read of the previous step must have COMPLETED before the barrier: the DMA issued right after it
        // refills that stage.  s_barrier alone is not a memory fence to the compiler -- with loop-invariant fragment
        // addresses it scheduled the last ds_reads of a step above the barrier and their lgkmcnt wait below it (a
        // run-to-run race).  The asm is a compiler barrier for memory operations and the wait itself.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt<(NST - 2) * LPS>();
        __builtin_amdgcn_s_barrier();                  // everyone's pieces of this step landed; everyone left the stage refilled next
        YV1_ISSUE((c + NST - 1) % NST);
        YV1_MFMA_BLOCK(smem + c * STAGE);
      }
    }
  }
  // remaining steps (fewer than NST that still issue, then the NST-1 that only drain): runtime stage index
  int cur = 0, nxt = NST - 1;                          // kt is a multiple of NST here
  for (; kt < nk; ++kt) {
    // retire K-step kt: the younger in-flight steps (at most NST-2 of them) may stay outstanding
    const int younger = min(nk - 1 - kt, NST - 2);
    // fragment reads of the previous step (the last steady-state step or the previous tail step) complete BEFORE the
    // barrier after which stage nxt is refilled -- same fence, same place as in the steady state above
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (younger >= 2) wait_vmcnt<2 * LPS>();
    else if (younger == 1) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                      // everyone's pieces of step kt landed; everyone left stage nxt
    if (kt + NST - 1 < nk && !(a.dbg & 1)) YV1_ISSUE(nxt);
    if (!(a.dbg & 2)) YV1_MFMA_BLOCK(smem + cur * STAGE);
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }
#undef YV1_MFMA_BLOCK
#undef YV1_SET_TAP_D
#undef YV1_ISSUE
  __syncthreads();                                     // all fragment reads done before the epilogue tile reuses LDS
  conv_epilogue<BM, BN, WM, WN, DB, true>(acc, a, smem, m0, n0, mt);
}

// ---- 3x3 stride-1 convolutions: the three taps of a filter row share ONE A tile (round 3) -------------------------
// The ring kernels fetch the A tile of every filter tap from L2 again: nine fetches of (almost) the same pixels per
// channel block.  These loops are bound by the global -> LDS fill rate, and on the 3x3 layers A is
// 1/2 (Cout 64-128) to 4/5 (DenseNet's 128 -> 32 growth convolutions) of what they fill.  For stride 1 / pad 1 the taps
// (r, 0), (r, 1), (r, 2) of pixel m read pixels m-1, m, m+1 of the flattened image (shifted by (r-1) rows): this kernel
// stages ONE A tile of BM + 2 consecutive pixels per (filter row, channel block) and runs the three taps off it, reading
// the fragments of tap s at LDS row m + s.  A-fill per MAC drops 3x; the weights (three tap tiles per stage) are as before.
//   * LDS row j of a stage holds the pixel with linear index m0 - 1 + j shifted by (r-1) image rows; it is zero (zero page)
//     when that pixel lies above / below the image.  A consumer m reading row m + s gets its horizontal neighbour
//     x + s - 1 -- unless that neighbour is outside the row (x == 0 with s == 0, x == W-1 with s == 2): those fragments
//     are zeroed per lane (the LDS row then holds the wrapped-around pixel of the adjacent image row).
//   * BM + 2 rows = BM/RPP whole DMA passes + one 1-KB piece issued by wave 0 alone; its counted vmcnt waits are one
//     deeper per stage than the other waves'.
//   * dgrad of the same geometry is the forward of the flipped filter: `flip` maps loop tap (r, s) to weight tap
//     (2-r, 2-s) of the transposed copy.
// One tile per workgroup, k_conv_dma's ring and the shared epilogue.
template <int BM, int BN, int BK, int WM, int WN, int NST, bool DB = false>
__global__ void __launch_bounds__(256, ((BM + 64 / (BK / 8)) * BK * 2 + 3 * BN * BK * 2) * NST <= 53 * 1024 ? 3 : (((BM + 64 / (BK / 8)) * BK * 2 + 3 * BN * BK * 2) * NST <= 80 * 1024 ? 2 : 1))
k_conv_h3(ConvArgs a, int flip) {
  constexpr int NTH = 256;
  static_assert(WM * WN == 4, "four waves");
  constexpr int CPR = BK / 8;
  constexpr int RPP = NTH / CPR;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tiles must be whole passes");
  constexpr int A_FULL = BM / RPP;                       // whole A passes (all waves)
  constexpr int XROWS = 64 / CPR;                        // rows of wave 0's extra 1-KB piece (>= 2)
  constexpr int A_ROWS = BM + XROWS;
  constexpr int BP1 = BN / RPP, B_PASSES = 3 * BP1;
  constexpr int LPS = A_FULL + B_PASSES;                 // DMA instructions per stage and wave (wave 0: + 1)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = A_ROWS * BK * 2, B1_BYTES = BN * BK * 2, STAGE = A_BYTES + 3 * B1_BYTES;
  static_assert((NST - 1) * (LPS + 1) < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  int mt, nt;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    mt = lin / a.NT;
    nt = lin - mt * a.NT;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int W = a.IW, H = a.IH;

  const int slot = tid % CPR, rrow = tid / CPR;
  const int lchunk = swz<BK>(rrow, slot);                // same key in every pass (RPP, BM are multiples of the key period)
  // centre pixel (linear index) and image row of the LDS rows this lane fills; pass A_FULL = wave 0's extra piece
  int pcs[A_FULL + 1], ys[A_FULL + 1];
#pragma unroll
  for (int i = 0; i <= A_FULL; ++i) {
    const int j = (i < A_FULL) ? rrow + i * RPP : BM + rrow;
    const int pc = m0 - 1 + j;
    pcs[i] = (pc >= 0 && pc < a.M) ? pc : -1;
    ys[i] = pc >= 0 ? (pc / W) % H : 0;
  }
  const int cblocks = a.Cin / BK;
  const int nk = 3 * cblocks;
  const int piece_row0 = (wid * 64) / CPR;

  int ld_r = 0, ld_cb = 0;
  const bf16_t* asrc[A_FULL + 1];
  int astep[A_FULL + 1];
  const bf16_t* wsrc[B_PASSES];
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_zero_page);
#define YV1_SET_ROW_H()                                                                                          \
  {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i <= A_FULL; ++i) {                                                        \
      const int yy = ys[i] + ld_r - 1;                                                                           \
      const bool ok = pcs[i] >= 0 && yy >= 0 && yy < H;                                                          \
      asrc[i] = ok ? a.X + ((size_t)(pcs[i] + (ld_r - 1) * W) * a.ldx + lchunk * 8) : zsrc;                      \
      astep[i] = ok ? BK : 0;                                                                                    \
    }                                                                                                            \
    const int wr = flip ? 2 - ld_r : ld_r;                                                                       \
    _Pragma("unroll") for (int b = 0; b < B_PASSES; ++b) {                                                       \
      const int s_ = b / BP1, bi_ = b - s_ * BP1;                                                                \
      const int ws = flip ? 2 - s_ : s_;                                                                         \
      wsrc[b] = a.W + ((size_t)(n0 + rrow + bi_ * RPP) * a.Kw + (wr * 3 + ws) * a.Cin + lchunk * 8);            \
    }                                                                                                            \
  }
#define YV1_ISSUE_H(STG_)                                                                                        \
  {                                                                                                              \
    unsigned char* sa_ = smem + (STG_) * STAGE;                                                                  \
    unsigned char* sb_ = sa_ + A_BYTES;                                                                          \
    _Pragma("unroll") for (int i = 0; i < A_FULL; ++i) {                                                         \
      __builtin_amdgcn_global_load_lds((glb_void*)asrc[i], (lds_void*)(sa_ + (piece_row0 + i * RPP) * (BK * 2)), 16, 0, 0); \
      asrc[i] += astep[i];                                                                                       \
    }                                                                                                            \
    if (wid == 0) {                                                                                              \
      __builtin_amdgcn_global_load_lds((glb_void*)asrc[A_FULL], (lds_void*)(sa_ + BM * (BK * 2)), 16, 0, 0);    \
      asrc[A_FULL] += astep[A_FULL];                                                                             \
    }                                                                                                            \
    _Pragma("unroll") for (int b = 0; b < B_PASSES; ++b) {                                                       \
      const int s_ = b / BP1, bi_ = b - s_ * BP1;                                                                \
      __builtin_amdgcn_global_load_lds((glb_void*)wsrc[b],                                                       \
                                       (lds_void*)(sb_ + s_ * B1_BYTES + (piece_row0 + bi_ * RPP) * (BK * 2)), 16, 0, 0); \
      wsrc[b] += BK;                                                                                             \
    }                                                                                                            \
    if (++ld_cb == cblocks) {                                                                                    \
      ld_cb = 0;                                                                                                 \
      ++ld_r;                                                                                                    \
      if (ld_r < 3) YV1_SET_ROW_H();                                                                             \
    }                                                                                                            \
  }
// counted wait: leave the DMA pieces of `YOUNGER_` whole stages in flight (wave 0 issues one more piece per stage)
#define YV1_WAIT_H(YOUNGER_)                                                                                     \
  {                                                                                                              \
    if (wid == 0) wait_vmcnt<(YOUNGER_) * (LPS + 1)>(); else wait_vmcnt<(YOUNGER_) * LPS>();                     \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  YV1_SET_ROW_H();
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < nk) YV1_ISSUE_H(p);

  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int KS = BK / 16;
  int fa_off[TM][3], fb_off[TN];
  bool mleft[TM], mright[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mrow = wm * (BM / WM) + i * 32 + l31;
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      const int row = mrow + s_;
      fa_off[i][s_] = row * (BK * 2) + swz<BK>(row, lh) * 16;
    }
    const int x = (m0 + mrow) % W;
    mleft[i] = x == 0;
    mright[i] = x == W - 1;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * (BN / WN) + j * 32 + l31;
    fb_off[j] = A_BYTES + row * (BK * 2) + swz<BK>(row, lh) * 16;
  }
#define YV1_MFMA_BLOCK_H(BASE_)                                                                                  \
  _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) {                                                             \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                          \
      bf16x8 fa[TM], fb[TN];                                                                                     \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>((BASE_) + (fa_off[i][s_] ^ (ks << 5))); \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>((BASE_) + s_ * B1_BYTES + (fb_off[j] ^ (ks << 5))); \
      if (s_ != 1) {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                         \
          const bool kill = s_ == 0 ? mleft[i] : mright[i];                                                      \
          u32x4 v = __builtin_bit_cast(u32x4, fa[i]);                                                            \
          const u32x4 z = {0u, 0u, 0u, 0u};                                                                      \
          v = kill ? z : v;                                                                                      \
          fa[i] = __builtin_bit_cast(bf16x8, v);                                                                 \
        }                                                                                                        \
      }                                                                                                          \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                           \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);                 \
    }                                                                                                            \
  }

  int kt = 0;
  {
    const int n_main = nk - (NST - 1);
    for (; kt + NST <= n_main; kt += NST) {
#pragma unroll
      for (int c = 0; c < NST; ++c) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // fragment reads of the previous stage complete (see k_conv_dma)
        YV1_WAIT_H(NST - 2);
        __builtin_amdgcn_s_barrier();
        YV1_ISSUE_H((c + NST - 1) % NST);
        YV1_MFMA_BLOCK_H(smem + c * STAGE);
      }
    }
  }
  int cur = 0, nxt = NST - 1;                            // kt is a multiple of NST here
  for (; kt < nk; ++kt) {
    const int younger = min(nk - 1 - kt, NST - 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (younger >= 2) { YV1_WAIT_H(2); }
    else if (younger == 1) { YV1_WAIT_H(1); }
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < nk) YV1_ISSUE_H(nxt);
    YV1_MFMA_BLOCK_H(smem + cur * STAGE);
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }
#undef YV1_MFMA_BLOCK_H
#undef YV1_SET_ROW_H
#undef YV1_ISSUE_H
#undef YV1_WAIT_H
  __syncthreads();
  conv_epilogue<BM, BN, WM, WN, DB>(acc, a, smem, m0, n0, mt);
}

// off ^ (ks << 5), computed where it is used: volatile asm so that the compiler does not hoist the KS variants of every
// fragment offset out of the K loop into registers (slice 0 needs no instruction)
__device__ __forceinline__ int frag_off(int off0, int ks) {
  if (ks == 0) return off0;
  int o;
  asm volatile("v_xor_b32 %0, %1, %2" : "=v"(o) : "v"(ks << 5), "v"(off0));
  return o;
}

// ---- persistent form of k_conv_dma ---------------------------------------------------------------------------
// Same tiles, ring, swizzle and MFMA loop, but a workgroup owns ONE column tile (nt) and walks the pixel tiles
// slot, slot + slots, slot + 2 slots ... of it:
//   * the first NST-1 K-steps of the NEXT pixel tile are issued (LDS-DMA) before the epilogue of the current one, so the
//     global-load latency of a tile is spent under the previous tile's epilogue instead of in front of its first MFMA
//     (the small-K 1x1 layers are load -> 16 MFMAs -> store: their tile lifetime was latency, not work);
//   * the epilogue has NO workgroup barrier: each wave packs its own sub-tile through a wave-private staging area --
//     16 pixel rows at a time, inside the ring stage the prefetch leaves free (stage NST-1) -- and stores full 128-B+
//     channel runs; waves drift apart across tiles instead of meeting three times per tile;
//   * BatchNorm statistic partials are accumulated in registers over all the tiles of the workgroup and written ONCE:
//     `slots` partial rows per launch instead of one per pixel tile (6272 -> 384 rows on the 112x112 maps), which is
//     what the finalize kernels then read;
//   * the epilogue's stores stay in flight across the next tile's first K-steps: the counted vmcnt of those steps
//     includes them (VM operations retire in issue order: the prefetch is OLDER than the stores, so waiting for it
//     does not wait for them).
// workgroups of k_conv_ps<...> a CU holds: by LDS (160 KB per CU), at most 3 (2 for the 128x256 tile: 128 accumulator VGPRs)
template <int BM, int BN, int BK, int NST>
constexpr int ps_wgs_per_cu() {
  constexpr int lds = NST * (BM + BN) * BK * 2;
  constexpr int by_lds = (160 * 1024) / lds;
  // VGPR budget per lane: 128 / 168 / 256 for the four-wave tiles; the eight-wave 256-row tiles (two waves per SIMD and
  // workgroup) get 256 (256x256: 128 accumulators) or 128 (256x128: 64 accumulators, two workgroups per CU)
  constexpr int cap = BM == 256 ? 1 : (BM * BN <= 128 * 64 ? 4 : (BM * BN <= 128 * 128 ? 3 : 2));
  return by_lds < 1 ? 1 : (by_lds < cap ? by_lds : cap);
}

// PLAIN: the training launches -- raw bf16 output in pixel order (+ optional statistics), no scale/shift/ReLU, no
// residual, no accumulate / shortcut-gradient reads.  Its loop contains no ordinary global load, so nothing makes the
// compiler drain the LDS-DMA queue (hipcc waits vmcnt(0) for any VGPR-destination load while a DMA is in flight) and the
// prefetch really flies under the epilogue; the generic form is correct with any epilogue but drains there.
template <int BM, int BN, int BK, int WM, int WN, int NST, bool PLAIN>
__global__ void __launch_bounds__(WM * WN * 64, (ps_wgs_per_cu<BM, BN, BK, NST>() * (WM * WN / 4))) k_conv_ps(ConvArgs a) {   // 2nd argument: waves per SIMD
  // Ping-pong main loop for the eight-wave tiles (below): measured SLOWER than the one-barrier loop on every layer it was
  // tried on (256->256 3x3 @28: 72.9 vs 69.2 us; four stages instead of three: no gain either) -- these loops are not
  // bound by the matrix pipe idling at the barrier.  Kept behind -DYV1_CONV_PP=1 as the record of that experiment.
#ifndef YV1_CONV_PP
#define YV1_CONV_PP 0
#endif
  constexpr bool PP = YV1_CONV_PP && WM * WN == 8 && NST >= 3;
  constexpr int NTH = WM * WN * 64;
  constexpr int CPR = BK / 8;
  constexpr int RPP = NTH / CPR;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tiles must be whole passes");
  constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP, LPS = A_PASSES + B_PASSES;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int CW = BN / WN;                                  // channels per wave
  constexpr int CPW = CW / 8;                                  // 16-B chunks per staged row
  constexpr int SPITCH = (CW / 2 % 32 == 16) ? CW * 2 : CW * 2 + 64;   // bytes; pitch/4 % 32 == 16: conflict-free ds_write_b32
  constexpr int SROWS = 16;                                    // rows staged at a time (half of a 32x32 block)
  constexpr int SPASSES = SROWS * CPW / 64;                    // 16-B stores per lane and staged half
  static_assert(SROWS * CPW % 64 == 0 && WM * WN * SROWS * SPITCH <= STAGE, "staging area must fit the free ring stage");
  constexpr int ESTORES = TM * 2 * SPASSES;                    // store instructions a wave issues per (full) tile
  static_assert(2 * LPS + ESTORES < 64 && NST >= 2, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  int nt, slot;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    slot = lin / a.NT;                                         // the column tiles of one pixel tile run side by side on one XCD
    nt = lin - slot * a.NT;
  }
  const int n0 = nt * BN;
  const int S = a.slots;
  if (a.stagger) {
    const int g = blockIdx.x / a.ncu;
    for (int i = 0; i < g * a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }

  const int cslot = tid % CPR, rrow = tid / CPR;
  const int lchunk = swz<BK>(rrow, cslot);
  int pix_base[A_PASSES], ph[A_PASSES], qw[A_PASSES];
  const bool direct = a.R * a.S == 1 && a.ah == 1 && a.aw == 1 && a.ch == 0 && a.cw == 0 && a.log2d == 0 &&
                      a.P == a.IH && a.Q == a.IW;
  const int cblocks = a.Cin / BK;
  const int nk = a.R * a.S * cblocks;
  const int dmask = (1 << a.log2d) - 1;
  int woff[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) woff[i] = (n0 + rrow + i * RPP) * a.Kw + lchunk * 8;
  const int piece_row0 = (wid * 64) / CPR;

  // ---- loader: runs ahead of the MFMA loop, across tile boundaries
  int ld_mt = slot;                                            // pixel tile the loader is fetching
  int ld_r = 0, ld_s = 0, ld_cb = 0;
  const bf16_t* asrc[A_PASSES];
  int astep[A_PASSES];
  const bf16_t* wsrc[B_PASSES];
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_zero_page);
#define YV1_SET_TAP_P()                                                                                          \
  {                                                                                                              \
    const int wtap_off = ((a.wr0 + ld_r * a.wrs) * a.WS + (a.ws0 + ld_s * a.wss)) * a.Cin;                       \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
      const int hn = ph[i] + ld_r * a.bh, wn_ = qw[i] + ld_s * a.bw;                                             \
      const int ih = hn >> a.log2d, iw = wn_ >> a.log2d;                                                         \
      const bool ok = pix_base[i] >= 0 && ((hn | wn_) & dmask) == 0 && hn >= 0 && wn_ >= 0 && ih < a.IH &&       \
                      iw < a.IW;                                                                                 \
      asrc[i] = ok ? a.X + ((size_t)(pix_base[i] + ih * a.IW + iw) * a.ldx + lchunk * 8) : zsrc;                 \
      astep[i] = ok ? BK : 0;                                                                                    \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i) wsrc[i] = a.W + (woff[i] + wtap_off);                   \
  }
#define YV1_SET_TILE_P()                                                                                         \
  {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
      const int m = ld_mt * BM + rrow + i * RPP;                                                                 \
      if (m < a.M) {                                                                                             \
        if (direct) {                                                                                            \
          pix_base[i] = m; ph[i] = 0; qw[i] = 0;                                                                 \
        } else {                                                                                                 \
          const int pq = a.P * a.Q;                                                                              \
          const int n = m / pq, rem = m - n * pq;                                                                \
          const int p = rem / a.Q, q = rem - p * a.Q;                                                            \
          pix_base[i] = n * a.IH * a.IW;                                                                         \
          ph[i] = p * a.ah + a.ch;                                                                               \
          qw[i] = q * a.aw + a.cw;                                                                               \
        }                                                                                                        \
      } else {                                                                                                   \
        pix_base[i] = -1; ph[i] = 0; qw[i] = 0;                                                                  \
      }                                                                                                          \
    }                                                                                                            \
    ld_r = 0; ld_s = 0; ld_cb = 0;                                                                               \
    YV1_SET_TAP_P();                                                                                             \
  }
#define YV1_ISSUE_P(STG_)                                                                                        \
  {                                                                                                              \
    unsigned char* sa_ = smem + (STG_) * STAGE;                                                                  \
    unsigned char* sb_ = sa_ + A_BYTES;                                                                          \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
      __builtin_amdgcn_global_load_lds((glb_void*)asrc[i], (lds_void*)(sa_ + (piece_row0 + i * RPP) * (BK * 2)), 16, 0, 0); \
      asrc[i] += astep[i];                                                                                       \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i) {                                                       \
      __builtin_amdgcn_global_load_lds((glb_void*)wsrc[i], (lds_void*)(sb_ + (piece_row0 + i * RPP) * (BK * 2)), 16, 0, 0); \
      wsrc[i] += BK;                                                                                             \
    }                                                                                                            \
    if (++ld_cb == cblocks) {                                                                                    \
      ld_cb = 0;                                                                                                 \
      if (++ld_s == a.S) { ld_s = 0; ++ld_r; }                                                                   \
      if (ld_r == a.R) {                                 /* last K-step of this tile: move on to the next tile */ \
        ld_mt += S;                                                                                              \
        if (ld_mt < a.MT) YV1_SET_TILE_P();                                                                      \
      } else {                                                                                                   \
        YV1_SET_TAP_P();                                                                                         \
      }                                                                                                          \
    }                                                                                                            \
  }

  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int KS = BK / 16;
  // fragment byte offsets of the first 16-deep slice; slice ks is the same offset with bits 5.. flipped by ks: the swizzled
  // chunk (2 ks + lh) ^ key(row) equals (lh ^ key) ^ 2 ks, so  off(ks) = off(0) ^ (ks << 5)  -- one v_xor per read instead
  // of KS registers per fragment row (what made the BK 64 forms spill or lose a wave per SIMD)
  int fa_off0[TM], fb_off0[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * (BM / WM) + i * 32 + l31;
    fa_off0[i] = row * (BK * 2) + swz<BK>(row, lh) * 16;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * (BN / WN) + j * 32 + l31;
    fb_off0[j] = A_BYTES + row * (BK * 2) + swz<BK>(row, lh) * 16;
  }
// One K-step.  The LDS-DMA of the step NST-1 ahead (ISSUE_: address VALU + ~60-180 issue cycles per instruction) goes
// AFTER the first fragment reads, so it executes while those ds_reads are in flight instead of in front of them.  The
// 256x256 tile (256-register budget) reads the fragments of BOTH 16-deep slices up front: its second slice's MFMAs then
// do not wait for their own LDS round trip; the four-wave tiles cannot afford those 24-32 extra registers (measured:
// spills or a lost wave per SIMD).
#define YV1_STEP_P(BASE_, ISSUE_)                                                                                \
  if constexpr (BM == 256 && BN == 256 && KS <= 2) {                                                             \
    bf16x8 fa[KS][TM], fb[KS][TN];                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                          \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>((BASE_) + frag_off(fa_off0[i], ks)); \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>((BASE_) + frag_off(fb_off0[j], ks)); \
    }                                                                                                            \
    ISSUE_;                                                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                            \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                           \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);         \
  } else {                                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                          \
      bf16x8 fa[TM], fb[TN];                                                                                     \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>((BASE_) + frag_off(fa_off0[i], ks)); \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>((BASE_) + frag_off(fb_off0[j], ks)); \
      if (ks == 0) { ISSUE_; }                                                                                   \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                           \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);                 \
    }                                                                                                            \
  }

  // BatchNorm statistics of this workgroup's column tile, summed over all its pixel tiles (lane = channel, see below)
  float st_s[TN], st_ss[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { st_s[j] = 0.f; st_ss[j] = 0.f; }

  // wave-private staging area inside ring stage NST-1, and this lane's place in the store passes
  unsigned char* stg = smem + (NST - 1) * STAGE + wid * (SROWS * SPITCH);
  const bool odd = lane & 1;
  constexpr int SSTEP = (64 / CPW) * SPITCH;                   // bytes between a lane's rows of consecutive store passes
  static_assert(3 * SSTEP < 65536, "ds_read offset field");
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned stg_rd = (unsigned)(unsigned long)(lds_u8*)(stg + (lane / CPW) * SPITCH + (lane % CPW) * 16);   // LDS byte address

  YV1_SET_TILE_P();
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < nk) YV1_ISSUE_P(p);
  int after_epilogue = 0;                                      // 1: ESTORES stores of the previous tile are younger than the prefetch

  for (int mt = slot; mt < a.MT; mt += S) {
    const int m0 = mt * BM;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if constexpr (PP) {
      // ---- ping-pong main loop (eight waves = two groups of four; waves w and w+4 share a SIMD) -------------------
      // A K-step is two phases separated by barriers:  R = fragment reads of the step + LDS-DMA issue of the step NST-1
      // ahead,  M = its 16 MFMAs.  Group 1 runs ONE barrier behind group 0 (an extra s_barrier on entry, one for group 0
      // on exit), so on every SIMD one wave is in M while the other is in R: the matrix pipe always has a wave with its
      // operands in registers, and the ds_reads / DMA issue / barrier latency of a step hide under the other group's
      // MFMAs (the one-barrier loop below idles the pipe for all of that, every step, on every SIMD at once).
      // Rules that keep the staggered groups safe (each wave, before the barrier X that ends ITS phase R of step k):
      //   * s_waitcnt lgkmcnt(0): its reads of stage k are complete -> when the partner barrier releases the other
      //     group into its next phase R, the stage that phase refills (the one read a step ago) is quiescent;
      //   * counted vmcnt for ITS pieces of step k+1: the other group's pass of this barrier is its entry to reading
      //     step k+1 (group 0), or the step is read one barrier later (group 1) -- either way everyone's pieces landed.
      const int grp = wid >> 2;
      {
        const int younger = min(NST - 2, nk - 1);              // prefetched steps behind step 0
        if (after_epilogue) {
          if (younger >= 2) wait_vmcnt<2 * LPS + ESTORES>(); else if (younger == 1) wait_vmcnt<LPS + ESTORES>(); else wait_vmcnt<ESTORES>();
        } else {
          if (younger >= 2) wait_vmcnt<2 * LPS>(); else if (younger == 1) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // staging reads of the previous epilogue
      if (grp == 1) __builtin_amdgcn_s_barrier();
      int cur = 0, nxt = NST - 1;
      for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();                          // Y: stage cur landed for everyone, stage nxt quiescent
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* base = smem + cur * STAGE;
        bf16x8 fa[KS][TM], fb[KS][TN];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(base + frag_off(fa_off0[i], ks));
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(base + frag_off(fb_off0[j], ks));
        }
        if (kt + NST - 1 < nk) YV1_ISSUE_P(nxt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads complete before X (and cannot sink below it)
        if (kt + 1 < nk) {
          const int younger = min(NST - 2, nk - 2 - kt);       // issued steps behind step kt+1
          if (after_epilogue && kt + 1 <= NST - 2) {           // step kt+1 was prefetched before the epilogue's stores
            if (younger >= 2) wait_vmcnt<2 * LPS + ESTORES>(); else if (younger == 1) wait_vmcnt<LPS + ESTORES>(); else wait_vmcnt<ESTORES>();
          } else {
            if (younger >= 2) wait_vmcnt<2 * LPS>(); else if (younger == 1) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
          }
        }
        __builtin_amdgcn_s_barrier();                          // X
        __builtin_amdgcn_sched_barrier(0);                     // keep the MFMAs below it
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
      }
      // group 0's closing barrier pairs with group 1's last X: every read of every stage is complete, every DMA of this
      // tile has been waited for -- the ring is free for the next tile's prefetch (group 1 still runs its last MFMAs)
      if (grp == 0) __builtin_amdgcn_s_barrier();
    } else {
      int kt = 0;
      {
        const int n_main = nk - (NST - 1);
        for (; kt + NST <= n_main; kt += NST) {
#pragma unroll
          for (int c = 0; c < NST; ++c) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads / staging reads complete before the barrier
            // the first NST-1 steps after an epilogue: its stores are younger than the step waited for, older than the
            // steps issued since -- leave them in flight
            if (kt == 0 && c < NST - 1 && after_epilogue) wait_vmcnt<(NST - 2) * LPS + ESTORES>();
            else wait_vmcnt<(NST - 2) * LPS>();
            __builtin_amdgcn_s_barrier();
            YV1_STEP_P(smem + c * STAGE, YV1_ISSUE_P((c + NST - 1) % NST));
          }
        }
      }
      int cur = 0, nxt = NST - 1;
      for (; kt < nk; ++kt) {
        const int younger = min(nk - 1 - kt, NST - 2);           // K-steps of THIS tile in flight behind step kt
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (after_epilogue && kt < NST - 1) {                    // step kt was prefetched before the epilogue: its stores are younger
          if (younger >= 2) wait_vmcnt<2 * LPS + ESTORES>();
          else if (younger == 1) wait_vmcnt<LPS + ESTORES>();
          else wait_vmcnt<ESTORES>();
        } else {
          if (younger >= 2) wait_vmcnt<2 * LPS>();
          else if (younger == 1) wait_vmcnt<LPS>();
          else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        YV1_STEP_P(smem + cur * STAGE, if (kt + NST - 1 < nk) YV1_ISSUE_P(nxt));
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
      }
      // every wave has finished reading the ring: the next tile's first K-steps may land in stages 0 .. NST-2 while the
      // epilogue runs; stage NST-1 is the staging area
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const bool more = mt + S < a.MT;
    if (more) {
#pragma unroll
      for (int p = 0; p < NST - 1; ++p)
        if (p < nk) YV1_ISSUE_P(p);
    }

    // ---- epilogue, per wave (no workgroup barrier)
    const bool full = m0 + BM <= a.M;
    if (a.stats) {
      // C/D layout of 32x32: col = lane&31 (channel), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel); rows past M are zero
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float v = acc[i][j][e];
            s += v;
            ss += v * v;
          }
        st_s[j] += s;
        st_ss[j] += ss;
      }
    }
    const bool affine = !PLAIN && (a.escale != nullptr || a.erelu);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        // accumulators -> bf16 rows in the staging area: lanes swap one value with their neighbour (DPP) so that each owns
        // a channel pair of one row, rounded by v_cvt_pk_bf16_f32
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = j * 32 + l31;                        // channel inside the wave's CW
          float al = 1.f, be = 0.f, lo_clamp = -3.0e38f;
          if constexpr (!PLAIN) {
            if (a.escale) { al = a.escale[n0 + wn * CW + col]; be = a.eshift[n0 + wn * CW + col]; }
            if (a.erelu && !a.ERES) lo_clamp = 0.f;
          }
#pragma unroll
          for (int e8 = 0; e8 < 8; e8 += 2) {
            const int e = hb * 8 + e8;
            const float mine_lo = affine ? fmaxf(acc[i][j][e] * al + be, lo_clamp) : acc[i][j][e];
            const float mine_hi = affine ? fmaxf(acc[i][j][e + 1] * al + be, lo_clamp) : acc[i][j][e + 1];
            const float send = odd ? mine_lo : mine_hi;
            const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xf, 0xf, true));
            const int lrow = ((e8 + (odd ? 1 : 0)) & 3) + 8 * (e8 >> 2) + 4 * lh;     // row inside the 16-row half
            const unsigned v = odd ? cvt_pk_bf16(recv, mine_hi) : cvt_pk_bf16(mine_lo, recv);
            *reinterpret_cast<unsigned*>(stg + lrow * SPITCH + (col & ~1) * 2) = v;
          }
        }
        // full-line stores of the 16 staged rows.  The same wave wrote them and LDS operations of a wave complete in order,
        // so no barrier; the reads are inline asm because hipcc would put s_waitcnt vmcnt(0) in front of a ds_read it
        // can see while an LDS-DMA is in flight (it cannot tell the staging area from the ring stages being filled) --
        // exactly the drain this kernel exists to avoid.  Loads and their wait are one statement (early-clobber outputs).
        u32x4 sv[SPASSES];
        if constexpr (SPASSES == 1) {
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(sv[0]) : "v"(stg_rd) : "memory");
        } else if constexpr (SPASSES == 2) {
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(sv[0]), "=&v"(sv[1]) : "v"(stg_rd), "i"(SSTEP) : "memory");
        } else {
          static_assert(SPASSES == 4, "staging passes");
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\tds_read_b128 %2, %4 offset:%6\n\t"
                       "ds_read_b128 %3, %4 offset:%7\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(sv[0]), "=&v"(sv[1]), "=&v"(sv[2]), "=&v"(sv[3])
                       : "v"(stg_rd), "i"(SSTEP), "i"(2 * SSTEP), "i"(3 * SSTEP) : "memory");
        }
#pragma unroll
        for (int sp = 0; sp < SPASSES; ++sp) {
          const int idx = lane + sp * 64;
          const int row = idx / CPW, cc = idx - row * CPW;
          const int m = m0 + wm * (BM / WM) + i * 32 + hb * 16 + row;
          const int nch = n0 + wn * CW + cc * 8;
          uint4 v = make_uint4(sv[sp][0], sv[sp][1], sv[sp][2], sv[sp][3]);
          if (full || m < a.M) {
            size_t off;
            if (PLAIN || a.os == 1) {
              off = (size_t)m * a.ldy + nch;
            } else {
              const int pq = a.P * a.Q;
              const int n = m / pq, rem = m - n * pq;
              const int p = rem / a.Q, q = rem - p * a.Q;
              off = ((size_t)(n * a.OH + p * a.os + a.oh0) * a.OW + q * a.os + a.ow0) * a.ldy + nch;
            }
            if (!PLAIN && a.AS) {
              const uint4 o = *reinterpret_cast<const uint4*>(a.AS + (size_t)m * a.ldas + nch);
              const unsigned mb = a.AM[(size_t)m * a.ldam + (nch >> 3)];
              const unsigned* pv = reinterpret_cast<const unsigned*>(&v);
              const unsigned* po = reinterpret_cast<const unsigned*>(&o);
              unsigned res[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float lo = __uint_as_float(pv[k] << 16) + (((mb >> (2 * k)) & 1u) ? __uint_as_float(po[k] << 16) : 0.f);
                const float hi = __uint_as_float(pv[k] & 0xffff0000u) +
                                 (((mb >> (2 * k + 1)) & 1u) ? __uint_as_float(po[k] & 0xffff0000u) : 0.f);
                res[k] = pack_bf16x2(lo, hi);
              }
              v = make_uint4(res[0], res[1], res[2], res[3]);
            }
            if (!PLAIN && a.ERES) {
              const uint4 o = *reinterpret_cast<const uint4*>(a.ERES + (size_t)m * a.ldres + nch);
              const unsigned* pv = reinterpret_cast<const unsigned*>(&v);
              const unsigned* po = reinterpret_cast<const unsigned*>(&o);
              const float lo_clamp = a.erelu ? 0.f : -3.0e38f;
              unsigned res[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float lo = fmaxf(__uint_as_float(pv[k] << 16) + __uint_as_float(po[k] << 16), lo_clamp);
                const float hi = fmaxf(__uint_as_float(pv[k] & 0xffff0000u) + __uint_as_float(po[k] & 0xffff0000u), lo_clamp);
                res[k] = pack_bf16x2(lo, hi);
              }
              v = make_uint4(res[0], res[1], res[2], res[3]);
            }
            if (!PLAIN && a.accumulate) {
              const uint4 o = *reinterpret_cast<const uint4*>(a.Y + off);
              const unsigned* pv = reinterpret_cast<const unsigned*>(&v);
              const unsigned* po = reinterpret_cast<const unsigned*>(&o);
              unsigned res[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float lo = __uint_as_float(pv[k] << 16) + __uint_as_float(po[k] << 16);
                const float hi = __uint_as_float(pv[k] & 0xffff0000u) + __uint_as_float(po[k] & 0xffff0000u);
                res[k] = pack_bf16x2(lo, hi);
              }
              v = make_uint4(res[0], res[1], res[2], res[3]);
            }
            *reinterpret_cast<uint4*>(a.Y + off) = v;
          }
        }
      }
    }
    // the counted waits of the next tile's first steps may skip ESTORES stores only if every wave certainly issued them
    after_epilogue = (more && full) ? 1 : 0;
  }
#undef YV1_STEP_P
#undef YV1_SET_TAP_P
#undef YV1_SET_TILE_P
#undef YV1_ISSUE_P

  if (a.stats) {
    // one partial row per workgroup: reduce the two half-waves, then the WM waves that share a column range
    __syncthreads();                                           // every wave is done with the ring and its staging area
    float* red = reinterpret_cast<float*>(smem);               // [WM][2][BN]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = st_s[j], ss = st_ss[j];
      s += __shfl_xor(s, 32, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (lh == 0) {
        const int c = wn * CW + j * 32 + l31;
        red[(wm * 2 + 0) * BN + c] = s;
        red[(wm * 2 + 1) * BN + c] = ss;
      }
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        s += red[(w * 2 + 0) * BN + tid];
        ss += red[(w * 2 + 1) * BN + tid];
      }
      float* o = a.stats + (size_t)slot * 2 * a.Cout + n0 + tid;
      o[0] = s;
      o[a.Cout] = ss;
    }
  }
}

int env_int(const char* name, int dflt);

// CUs of the current device (cached)
int ps_ncu() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
    else
      ncu = 256;                                               // MI355X
  }
  return ncu;
}

// Workgroups of one k_conv_ps instantiation that are resident per CU: the smaller of what its LDS request and its
// ALLOCATED registers admit (512 VGPRs per SIMD lane, granule 8; a 256-thread workgroup is one wave per SIMD).
// launch_bounds is only a hint to the register allocator -- the generic-epilogue forms exceed the 128-VGPR budget of four
// workgroups per CU -- so the grid is sized from the compiled kernel, not from the hint: a persistent grid larger than
// what is resident would run its surplus workgroups as a second, mostly idle round.
template <int BM, int BN, int BK, int WM, int WN, int NST, bool PLAIN>
int ps_resident_per_cu() {
  static int cached = 0;
  if (cached) return cached;
  constexpr int lds = NST * (BM + BN) * BK * 2;
  int by_lds = (160 * 1024) / lds;
  if (by_lds < 1) by_lds = 1;
  int by_regs = ps_wgs_per_cu<BM, BN, BK, NST>();
  hipFuncAttributes attr;
  if (hipFuncGetAttributes(&attr, (const void*)k_conv_ps<BM, BN, BK, WM, WN, NST, PLAIN>) == hipSuccess && attr.numRegs > 0) {
    const int alloc = (attr.numRegs + 7) / 8 * 8;
    by_regs = (512 / alloc) / (WM * WN / 4);                   // waves per SIMD / waves per SIMD of one workgroup
    if (by_regs > 8) by_regs = 8;
    if (by_regs < 1) by_regs = 1;
  }
  int r = by_lds < by_regs ? by_lds : by_regs;
  if (r > 4) r = 4;                                            // more than 16 waves per CU buys nothing here
  cached = r;
  return r;
}

// pixel-tile slots per column tile for a persistent launch: as many workgroups as the chip holds, at most one per tile
int ps_slots(int MT, int NT, int per_cu) {
  static int reserve = -1;                                     // tuning: leave this many workgroup slots per CU to other streams
  if (reserve < 0) reserve = env_int("YV1_PS_RESERVE", 0);
  if (per_cu - reserve >= 1) per_cu -= reserve;
  int s = ps_ncu() * per_cu / NT;
  if (s < 1) s = 1;
  if (s >= MT) return MT;
  static int balance = -1;
  if (balance < 0) balance = env_int("YV1_PS_BALANCE", 0);
  if (!balance) return s;
  // balanced variant (measured: slower -- fewer workgroups in flight cost more than the ragged last round)
  const int k = (MT + s - 1) / s;
  return (MT + k - 1) / k;
}

template <int BM, int BN, int BK, int WM, int WN, int NST>
int launch_ps(ConvArgs& a, hipStream_t stream) {
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr size_t LDS = (size_t)NST * STAGE > (size_t)WM * 2 * BN * 4 ? (size_t)NST * STAGE : (size_t)WM * 2 * BN * 4;
  a.MT = (a.M + BM - 1) / BM;
  a.NT = a.Cout / BN;
    const bool plain = !a.escale && !a.erelu && !a.AS && !a.ERES && !a.accumulate && a.os == 1;
  {
    static int stagger = -1;
    if (stagger < 0) stagger = env_int("YV1_PS_STAGGER", 0);
    a.stagger = stagger; a.ncu = ps_ncu();
  }
  a.slots = ps_slots(a.MT, a.NT, plain ? ps_resident_per_cu<BM, BN, BK, WM, WN, NST, true>()
                                       : ps_resident_per_cu<BM, BN, BK, WM, WN, NST, false>());
  {
    const bool direct = a.R * a.S == 1 && a.ah == 1 && a.aw == 1 && a.ch == 0 && a.cw == 0 && a.log2d == 0 &&
                        a.P == a.IH && a.Q == a.IW;
    yv1_cfg_note("k_conv_ps<%d,%d,%d,%d,%d,%d,%s>%s", BM, BN, BK, WM, WN, NST, plain ? "plain" : "generic", direct ? " direct" : "");
  }
  if (plain) {
    auto kern = k_conv_ps<BM, BN, BK, WM, WN, NST, true>;
    if (LDS > 64 * 1024) YV1_SET_MAX_LDS(kern, LDS);
    hipLaunchKernelGGL(kern, dim3(a.slots * a.NT), dim3(WM * WN * 64), LDS, stream, a);
  } else {
    auto kern = k_conv_ps<BM, BN, BK, WM, WN, NST, false>;
    if (LDS > 64 * 1024) YV1_SET_MAX_LDS(kern, LDS);
    hipLaunchKernelGGL(kern, dim3(a.slots * a.NT), dim3(WM * WN * 64), LDS, stream, a);
  }
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

template <int BM, int BN, int BK, int WM, int WN, int NST, bool DB = false>
int launch_dma(ConvArgs& a, hipStream_t stream) {
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int EPI_PITCH = (BN / 2 % 32 == 16) ? BN * 2 : BN * 2 + 64;
  constexpr int EPI = BM * EPI_PITCH + WM * 2 * BN * 4 + (DB ? 5 * BN * 4 : 0);     // DB: + the per-channel vectors
  constexpr size_t LDS = NST * STAGE > EPI ? NST * STAGE : EPI;
  static_assert(!DB || LDS >= (size_t)2 * 8 * WM * WN * 64 * 4, "the deferred-BatchNorm sums reduce through 2 x [RG][BN] floats");
  static_assert(!DB || LDS >= (size_t)EPI, "the DB epilogue's per-channel vectors sit behind the statistics area");
  a.MT = (a.M + BM - 1) / BM;
  a.NT = DB ? (a.Cout + BN - 1) / BN : a.Cout / BN;       // DB: a partial last column tile (guarded epilogue, zero weight rows)
  auto kern = k_conv_dma<BM, BN, BK, WM, WN, NST, DB>;
  if (LDS > 64 * 1024) YV1_SET_MAX_LDS(kern, LDS);
  {
    const bool direct = a.R * a.S == 1 && a.ah == 1 && a.aw == 1 && a.ch == 0 && a.cw == 0 && a.log2d == 0 &&
                        a.P == a.IH && a.Q == a.IW;         // the kernel's own condition for its direct addressing
    yv1_cfg_note("k_conv_dma<%d,%d,%d,%d,%d,%d>%s%s", BM, BN, BK, WM, WN, NST, direct ? " direct" : "", DB ? (a.db_unit ? " bn-sums" : " bn-deferred") : "");
  }
  hipLaunchKernelGGL(kern, dim3(a.MT * a.NT), dim3(WM * WN * 64), LDS, stream, a);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

template <int BM, int BN, int BK, int WM, int WN, int NST, bool DB = false>
int launch_h3(ConvArgs& a, hipStream_t stream) {
  constexpr int CPR = BK / 8;
  constexpr int STAGE = (BM + 64 / CPR) * BK * 2 + 3 * BN * BK * 2;
  constexpr int EPI_PITCH = (BN / 2 % 32 == 16) ? BN * 2 : BN * 2 + 64;
  constexpr int EPI = BM * EPI_PITCH + WM * 2 * BN * 4;
  constexpr size_t LDS = NST * STAGE > EPI ? NST * STAGE : EPI;
  static_assert(!DB || LDS >= (size_t)2 * 8 * 256 * 4, "the BatchNorm-backward sums reduce through 2 x [RG][BN] floats");
  static_assert(!DB || LDS >= (size_t)EPI, "the DB epilogue's per-channel vectors sit behind the statistics area");
  a.MT = (a.M + BM - 1) / BM;
  a.NT = a.Cout / BN;
  auto kern = k_conv_h3<BM, BN, BK, WM, WN, NST, DB>;
  if (LDS > 64 * 1024) YV1_SET_MAX_LDS(kern, LDS);
  const int flip = a.bh < 0 ? 1 : 0;
  yv1_cfg_note("k_conv_h3<%d,%d,%d,%d,%d,%d>%s%s", BM, BN, BK, WM, WN, NST, flip ? " flipped" : "", DB ? " bn-sums" : "");
  hipLaunchKernelGGL(kern, dim3(a.MT * a.NT), dim3(256), LDS, stream, a, flip);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

template <int BM, int BN, int BK, int WM, int WN>
int launch(ConvArgs& a, hipStream_t stream) {
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int EPI_PITCH = (BN / 2 % 32 == 16) ? BN * 2 : BN * 2 + 64;
  constexpr int EPI = BM * EPI_PITCH + WM * 2 * BN * 4;
  a.MT = (a.M + BM - 1) / BM;
  a.NT = a.Cout / BN;
  // single-K-step problems (1x1 conv with Cin == BK) never touch the second staging buffer: a smaller LDS
  // request lets more workgroups share a CU, which is what these load->few-MFMA->store kernels need
  const int nk = a.R * a.S * (a.Cin / BK);
  const size_t stage_bytes = (size_t)(nk > 1 ? 2 : 1) * STAGE;
  const size_t lds = stage_bytes > (size_t)EPI ? stage_bytes : (size_t)EPI;
  auto kern = k_conv_gemm<BM, BN, BK, WM, WN>;
  constexpr size_t MAXLDS = 2 * STAGE > EPI ? 2 * STAGE : EPI;
  if (MAXLDS > 64 * 1024) YV1_SET_MAX_LDS(kern, MAXLDS);
  yv1_cfg_note("k_conv_gemm<%d,%d,%d,%d,%d>", BM, BN, BK, WM, WN);
  hipLaunchKernelGGL(kern, dim3(a.MT * a.NT), dim3(WM * WN * 64), lds, stream, a);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// tile choice shared by dispatch() and yv1_conv2d_stats_rows(): returns BM
int choose_cfg(int M, int Cout, int Cin, int* bn) {
  static int forced = -2;
  if (forced == -2) {                       // debug/tuning override: YV1_CONV_CFG=<BM>x<BN>
    const char* e = getenv("YV1_CONV_CFG");
    forced = -1;
    if (e) { int m = 0, n = 0; if (sscanf(e, "%dx%d", &m, &n) == 2) forced = m * 1000 + n; }
  }
  if (forced > 0) {
    const int fm = forced / 1000, fn = forced % 1000;
    if (Cout % fn == 0 && (Cin % 64) == 0) { *bn = fn; return fm; }
  }
  // measured per layer on MI355X (tools/bench_conv.py sweep): 128x128 wins whenever it yields >= ~200 tiles,
  // below that (7x7 feature maps) 64x64 fills the 256 CUs better
  const long long tiles128 = (long long)((M + 127) / 128) * ((Cout + 127) / 128);
  if (Cout % 128 == 0 && tiles128 >= 192) { *bn = 128; return 128; }
  if (Cout % 64 == 0) {
    const long long tiles = (long long)((M + 127) / 128) * (Cout / 64);
    *bn = 64;
    return tiles >= 512 ? 128 : 64;
  }
  *bn = 32;
  return 128;
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// The kernel a convolution GEMM of this shape runs: kind 0 = k_conv_gemm (register-staged), 1 = k_conv_dma (LDS-DMA
// ring, one tile per workgroup), 2 = k_conv_ps (LDS-DMA ring, persistent workgroups).  One function decides for
// dispatch() and for yv1_conv2d_stats_rows(), which must agree on the number of statistic partial rows.
struct ConvPlan { int kind, bm, bn, bk, nst; };

// 3x3, stride 1, pad 1, same-size output (forward, or the data gradient of such a convolution): k_conv_h3's geometry
bool is_s1p1_3x3(const ConvArgs& a) {
  return a.R == 3 && a.S == 3 && a.log2d == 0 && a.ah == 1 && a.aw == 1 && a.os == 1 && a.P == a.IH && a.Q == a.IW &&
         a.wr0 == 0 && a.ws0 == 0 && a.wrs == 1 && a.wss == 1 && a.WS == 3 && a.Kw == 9 * a.Cin &&
         ((a.bh == 1 && a.ch == -1 && a.bw == 1 && a.cw == -1) || (a.bh == -1 && a.ch == 1 && a.bw == -1 && a.cw == 1));
}

ConvPlan plan_conv(int M, int Cout, int Cin, int taps, bool s1p1 = false) {
  ConvPlan p;
  {
    // kind 3 = k_conv_h3 (the three taps of a filter row off one A tile).  YV1_CONV_H3=0 turns it off, YV1_CONV_H3_NST
    // forces its ring depth (tuning)
    static int h3 = -1, h3nst = 0, h3min = 0, h3bm = 0;
    if (h3 < 0) {
      h3 = env_int("YV1_CONV_H3", 1); h3nst = env_int("YV1_CONV_H3_NST", 0); h3min = env_int("YV1_CONV_H3_MIN_TILES", 192);
      h3bm = env_int("YV1_CONV_H3_BM", 0);
    }
    // measured per layer at batch 64 (tools/bench_h3.py, gpurun_out/r3e): forward 128 -> 32 (DenseNet growth) 168 -> 113 us
    // @112, 51 -> 36 @56, 19 -> 12 @28; 64 -> 64 @112 108 -> 94 (dgrad 126 -> 108) with two stages; 128 -> 128 @56 105 -> 93
    // (91 -> 85); level on 512 -> 512 @14; SLOWER where K per tap is one 32-channel block (the data gradient of the growth
    // convolutions: 28 -> 34 us @56) and against the 128x256 / 256x256 tiles of 256 -> 256 @28 (80 vs 83): those keep the ring
    // kernels.
    // (A weight-stationary persistent form for the two shapes whose filter bank fits in LDS -- 128 -> 32 and 64 -> 64:
    // 73.7 KB -- was built and measured: one workgroup per CU cannot hide its LDS round trips and dependent-MFMA latency;
    // 147-165 us against 113 here on 128 -> 32 @112 even with two fragment buffers and two accumulator sets.  Removed.)
    if (h3 && s1p1 && taps == 9 && Cin % 64 == 0 && !(Cout == 256 && M <= 60000)) {
      int bm = 128, bn = 0, bk = 32, nst = 2;
      if (Cout % 128 == 0) { bn = 128; }
      else if (Cout % 64 == 0) { bn = 64; if (h3bm == 256) bm = 256; }
      else if (Cout % 32 == 0) { bn = 32; bk = 64; if (h3bm == 256) bm = 256; }
      if ((h3nst == 2 || h3nst == 3) && bm == 128) nst = h3nst;
      const long long tiles = (long long)((M + bm - 1) / bm) * (bn ? Cout / bn : 0);
      if (bn && tiles >= h3min) { p.kind = 3; p.bm = bm; p.bn = bn; p.bk = bk; p.nst = nst; return p; }
    }
  }
  static int dma = -1, fbk = -1, ps = -1;
  if (dma < 0) {
    dma = env_int("YV1_CONV_DMA", 1);       // 0: register-staged loop; 2/3/4: force a stage count
    fbk = env_int("YV1_CONV_BK", 0);        // 32 | 64: force the K-step
    ps = env_int("YV1_CONV_PS", 1);         // 0: one tile per workgroup (k_conv_dma) instead of persistent workgroups
  }
  p.bm = choose_cfg(M, Cout, Cin, &p.bn);
  const bool c64 = (Cin % 64) == 0;
  if (!dma || !(c64 || p.bn != 32) || (p.bm == 128 && p.bn == 32 && !c64)) {
    // register-staged loop.  BK: 64 halves the barriers per MAC but its 64 KB of LDS allows only 2 workgroups per CU;
    // measured per layer: BK 32 on the large, bandwidth-bound feature maps, BK 64 on the deep compute-bound layers
    bool k64 = c64 && M < 150000;
    if (fbk == 32) k64 = false;
    if (fbk == 64) k64 = c64;
    p.kind = 0; p.bk = k64 ? 64 : 32; p.nst = 2;
    return p;
  }
  // LDS-DMA ring.  Measured per layer (tools/bench_conv.py, N=64): BK 32 with three stages (48 KB, 3 workgroups/CU, two
  // K-steps in flight) wins on the bandwidth-bound feature maps; BK 64 with two stages (64 KB, 2 workgroups/CU) on the deep
  // 3x3 / Cin >= 1024 layers; the 64x64 tiles of the 7x7 maps take BK 64 with three stages.
  bool d64 = c64 && ((taps > 1 && M < 250000) || (Cin >= 1024 && M < 150000));
  int nst = d64 ? 2 : 3;
  if (p.bm == 64 && c64) { d64 = true; nst = 3; }
  if (fbk == 32) { d64 = false; if (dma == 1) nst = 3; }
  if (fbk == 64 && c64) { d64 = true; if (dma == 1) nst = (p.bm == 64) ? 3 : 2; }
  if (dma >= 2) nst = dma > 4 ? 4 : dma;
  if (d64 && nst > 3) nst = 3;
  p.kind = ps ? 2 : 1;
  // Persistent workgroups (k_conv_ps) are assigned their tiles statically; measured per layer at batch 64
  // (gpurun_out/r2f tables, same box, interleaved): 3-17 % faster wherever a tile is short (small K, wide output: the next
  // tile's loads fly under the epilogue) or compute-heavy, but 8-14 % slower on the 112x112 maps with <= 128 output
  // channels and on 512 -> 256 @56: those stream > 400 MB at the HBM rate already, their tiles take the same time with a
  // few workgroups running as with all of them, and 6272 (1568) tiles over 768-1024 (384) resident workgroups leave a
  // ragged last round that one-tile-per-workgroup launches fill dynamically.
  if (ps == 1 && ((M >= 600000 && Cout <= 128 && Cin * taps >= 256) || (M >= 150000 && M < 600000 && Cout == 256 && Cin >= 512 && taps == 1)))
    p.kind = 1;
  // one 256-wide column tile when it covers all of Cout: the gathered A rows are then fetched L2 -> LDS once
  // instead of twice (these loops are bound by that bandwidth); 9 % on 256->256 3x3 @28, 3 % on 1024->256
  if (dma == 1 && p.bm == 128 && p.bn == 128 && Cout == 256 && M <= 60000 && (taps > 1 || Cin >= 1024)) {
    // persistent: one 256x256 tile per CU, eight waves of 128x64 (0.75 fragment reads per MFMA instead of 1, weights
    // fetched once per 256 pixel rows), BK 32 in a three-stage ring (96 KB): measured +11-13 % over 128x256 on 256->256
    // 3x3 @28 and on its stride-2 sibling, +9 % on 1024->256 (four stages, BK 64 in two stages, and 256x128 / 128x256 eight-wave
    // tiles were all slower; 196 tiles on 256 CUs is what caps this shape)
    // Round 3: the default is chosen by STEP time, not by the kernel's own: alone the 256x256 tile wins (+11-13 %), inside
    // the training step it loses 0.3-0.7 % (2915-2932 vs 2907-2910 img/s interleaved, round 2: 2969-2972 vs 2951-2962) -- one
    // eight-wave workgroup per CU leaves the weight-gradient stream nothing to run beside.  YV1_CONV_T256=1 selects it
    // (the per-layer tables in profiles/ name which tile they were taken with).
    static int t256 = -1;
    if (t256 < 0) t256 = env_int("YV1_CONV_T256", 0);
    if (p.kind == 2 && t256 && M >= 256 * 128) { p.bm = 256; p.bn = 256; p.bk = 32; p.nst = 3; return p; }
    p.bn = 256; p.bk = 32; p.nst = 3;
    return p;
  }
  if (p.bm == 128 && p.bn == 32) { p.bk = 64; p.nst = 3; return p; }      // DenseNet growth convs (c64 here)
  if (p.bm == 256 && p.bn == 256) { p.bk = 32; p.nst = 3; return p; }                            // forced (YV1_CONV_CFG) only
  p.bk = d64 ? 64 : 32;
  p.nst = d64 ? (nst >= 3 ? 3 : 2) : (nst >= 4 ? 4 : (nst == 3 ? 3 : 2));
  if (p.kind == 2 && p.nst > 3) p.kind = 1;                                // four-stage rings: tuning builds only
  return p;
}

// tile of a deferred-BatchNorm data gradient (1x1, K = Cin a multiple of 64): choose_cfg's tile, the ring depth fixed per tile
// wt_rows: rows of the weight operand that may be read (>= Cout; zero beyond Cout).  Cout = 64j + 32 (every other DenseNet
// layer) would otherwise run 32-wide tiles -- 99 us per launch against 67-72 for the 64 / 128-wide ones, the gradient
// operand re-read Cout/32 times: with padded weights the last column tile simply reaches past Cout (guarded epilogue).
ConvPlan plan_deferred(int M, int Cout, int Cin, int wt_rows) {
  ConvPlan p;
  p.kind = 1;
  int cr = Cout;
  if (Cout % 64 == 32 && Cout > 32) {
    const int c128 = (Cout + 127) / 128 * 128, c64 = (Cout + 63) / 64 * 64;
    cr = (Cout % 128 == 96 && wt_rows >= c128) ? c128 : (wt_rows >= c64 ? c64 : Cout);
  }
  p.bm = choose_cfg(M, cr, Cin, &p.bn);
  p.bk = (p.bn == 32 || p.bm == 64) ? 64 : 32;
  p.nst = 3;
  return p;
}

bool db_two_stage() {
  static int v = -1;
  if (v < 0) v = env_int("YV1_DB_NST2", 1);               // tuning: 0 keeps the three-stage ring for the pointwise 128x64 DB form
  return v != 0;
}

// Kernel of a data gradient with the BatchNorm-backward epilogue: the pointwise form (plan_deferred), k_conv_h3 where the
// plain data gradient would run it (3x3 stride 1 with a 64-channel K block), else k_conv_dma with the 32-channel K step and
// the three-stage ring -- the DB instantiations.  kind 0: not available for this shape (the caller keeps the separate passes).
ConvPlan plan_db(int M, int Cout, int Cin, int taps, bool s1p1, int wt_rows) {
  if (taps == 1) return plan_deferred(M, Cout, Cin, wt_rows);
  ConvPlan q = plan_conv(M, Cout, Cin, taps, s1p1);
  ConvPlan p;
  p.kind = 0; p.bm = q.bm; p.bn = q.bn; p.bk = 32; p.nst = 3;
  if (q.kind == 3) {
    if (q.bm == 128 && q.bk == 32 && (q.bn == 128 || q.bn == 64)) { p.kind = 3; p.nst = 2; }
    return p;
  }
  if (q.bn == 256 || q.bm == 256) return p;              // the 256-wide / 256x256 tiles have no DB form
  if (Cin % 64 && q.bn == 32) return p;                  // 128x32 needs the 64-channel K step
  p.kind = 1;
  p.bk = (q.bn == 32 || q.bm == 64) ? 64 : 32;
  if (p.bk == 64 && Cin % 64) p.kind = 0;
  return p;
}

// every (tile, K-step, stage count) the ring kernels are instantiated for
#define YV1_RING_CASES                 \
  YV1_RING_CASE(128, 256, 32, 2, 2, 3) \
  YV1_RING_CASE(128, 32, 64, 4, 1, 3)  \
  YV1_RING_CASE(128, 128, 32, 2, 2, 3) \
  YV1_RING_CASE(128, 128, 32, 2, 2, 2) \
  YV1_RING_CASE(128, 128, 64, 2, 2, 2) \
  YV1_RING_CASE(128, 128, 64, 2, 2, 3) \
  YV1_RING_CASE(128, 64, 32, 2, 2, 3)  \
  YV1_RING_CASE(128, 64, 32, 2, 2, 2)  \
  YV1_RING_CASE(128, 64, 64, 2, 2, 2)  \
  YV1_RING_CASE(128, 64, 64, 2, 2, 3)  \
  YV1_RING_CASE(64, 64, 32, 2, 2, 3)   \
  YV1_RING_CASE(64, 64, 32, 2, 2, 2)   \
  YV1_RING_CASE(64, 64, 64, 2, 2, 2)   \
  YV1_RING_CASE(64, 64, 64, 2, 2, 3)   \
  YV1_RING_CASE(256, 256, 32, 2, 4, 3)

int dispatch(ConvArgs& a, hipStream_t stream) {
  if (a.Cin % 32 || a.Cout % 32 || a.ldx % 8 || a.ldy % 8) return YV1_ERR_UNSUPPORTED;
  {
    static int dbg = -1;
    if (dbg < 0) dbg = env_int("YV1_CONV_DBG", 0);
    a.dbg = dbg;
  }
  if (a.DBX) {
    // BatchNorm-backward epilogue (ConvArgs::DBX), one tile per workgroup: yv1_conv2d_dgrad_bn_deferred_rows() /
    // yv1_conv2d_dgrad_bn_sums_rows() = its pixel tiles.  k_conv_dma<..., DB = true> (any stride-1 geometry) or k_conv_h3<..., DB>
    if (a.os != 1 || a.AS || a.OM || a.gsum || a.X2 || a.stats || a.escale || a.ERES) return YV1_ERR_UNSUPPORTED;
    ConvPlan p = plan_db(a.M, a.Cout, a.Cin, a.R * a.S, is_s1p1_3x3(a), a.db_wt_rows);
    if (p.kind == 3) {
      if (p.bn == 128) return launch_h3<128, 128, 32, 2, 2, 2, true>(a, stream);
      if (p.bn == 64) return launch_h3<128, 64, 32, 2, 2, 2, true>(a, stream);
      return YV1_ERR_UNSUPPORTED;
    }
    if (p.kind != 1) return YV1_ERR_UNSUPPORTED;
    if (p.bm == 128 && p.bn == 128) return launch_dma<128, 128, 32, 2, 2, 3, true>(a, stream);
    // pointwise 128x64: a two-stage ring (K is one or two hundred channels: the tile is all epilogue) in 27 KB of LDS and
    // <= 102 VGPRs -- five workgroups per CU instead of four; the tile's latency chain, not bandwidth, bounds this kernel
    if (p.bm == 128 && p.bn == 64 && a.R * a.S == 1 && db_two_stage()) return launch_dma<128, 64, 32, 2, 2, 2, true>(a, stream);
    if (p.bm == 128 && p.bn == 64) return launch_dma<128, 64, 32, 2, 2, 3, true>(a, stream);
    if (p.bm == 64 && p.bn == 64) return launch_dma<64, 64, 64, 2, 2, 3, true>(a, stream);
    if (p.bm == 128 && p.bn == 32) return launch_dma<128, 32, 64, 4, 1, 3, true>(a, stream);
    return YV1_ERR_UNSUPPORTED;
  }
  ConvPlan p = plan_conv(a.M, a.Cout, a.Cin, a.R * a.S, is_s1p1_3x3(a));
  if (p.kind == 3) {
#define YV1_H3_CASE(BN_, BK_, WM_, WN_)                                                                          \
    if (p.bm == 128 && p.bn == BN_ && p.bk == BK_) return p.nst == 3 ? launch_h3<128, BN_, BK_, WM_, WN_, 3>(a, stream) \
                                                                     : launch_h3<128, BN_, BK_, WM_, WN_, 2>(a, stream);
    YV1_H3_CASE(128, 32, 2, 2) YV1_H3_CASE(64, 32, 2, 2) YV1_H3_CASE(32, 64, 4, 1)
#undef YV1_H3_CASE
    if (p.bm == 256 && p.bn == 64 && p.bk == 32) return launch_h3<256, 64, 32, 4, 1, 2>(a, stream);
    if (p.bm == 256 && p.bn == 32 && p.bk == 64) return launch_h3<256, 32, 64, 4, 1, 2>(a, stream);
    return YV1_ERR_UNSUPPORTED;
  }
  // The shortcut-adding dgrad (yv1_conv2d_dgrad_add_masked_nhwc_bf16) streams three 4p-wide tensors per tile through a
  // generic epilogue.  One tile per workgroup, whose epilogue fetches the shortcut operands of all its store passes up front,
  // beats the persistent form (which waits once per 16-row group): 198 vs 249 us at 112x112, 122 vs 148 us at 56x56,
  // 68 vs 72 us at 28x28, level below
  static int as_min_m = -1;
  if (as_min_m < 0) as_min_m = env_int("YV1_AS_DMA_MIN_M", 0);
  if (p.kind == 2 && a.AS && a.M >= as_min_m) p.kind = 1;
  if (p.kind == 2 && (a.X2 || a.OM || a.gsum)) p.kind = 1;       // second K source / output mask / column sums: k_conv_dma only
  if (p.kind == 0 && (a.X2 || a.OM || a.gsum)) return YV1_ERR_UNSUPPORTED;
  if (p.kind == 0) {
    const bool k64 = p.bk == 64;
    if (p.bm == 128 && p.bn == 128) return k64 ? launch<128, 128, 64, 2, 2>(a, stream) : launch<128, 128, 32, 2, 2>(a, stream);
    if (p.bm == 128 && p.bn == 64) return k64 ? launch<128, 64, 64, 2, 2>(a, stream) : launch<128, 64, 32, 2, 2>(a, stream);
    if (p.bm == 64 && p.bn == 64) return k64 ? launch<64, 64, 64, 2, 2>(a, stream) : launch<64, 64, 32, 2, 2>(a, stream);
    if (p.bm == 128 && p.bn == 32) return k64 ? launch<128, 32, 64, 4, 1>(a, stream) : launch<128, 32, 32, 4, 1>(a, stream);
    return YV1_ERR_UNSUPPORTED;
  }
#define YV1_RING_CASE(BM_, BN_, BK_, WM_, WN_, NST_)                                                             \
  if (p.bm == BM_ && p.bn == BN_ && p.bk == BK_ && p.nst == NST_)                                                \
    return p.kind == 2 ? launch_ps<BM_, BN_, BK_, WM_, WN_, NST_>(a, stream) : launch_dma<BM_, BN_, BK_, WM_, WN_, NST_>(a, stream);
  YV1_RING_CASES
#undef YV1_RING_CASE
  if (p.nst == 4) {                                                       // tuning builds (YV1_CONV_DMA=4)
    if (p.bm == 128 && p.bn == 128) return launch_dma<128, 128, 32, 2, 2, 4>(a, stream);
    if (p.bm == 128 && p.bn == 64) return launch_dma<128, 64, 32, 2, 2, 4>(a, stream);
    if (p.bm == 64 && p.bn == 64) return launch_dma<64, 64, 32, 2, 2, 4>(a, stream);
  }
  return YV1_ERR_UNSUPPORTED;
}

// NCHW fp32 image -> zero-padded NHWC4 bf16 [N][H+6][W+6][4] (pad 3 each side, channel 3 = 0).
__global__ void k_pack_input(const float* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W) {
  const int HP = H + 6, WP = W + 6;
  const long long total = (long long)N * HP * WP;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int wp = (int)(i % WP);
    const long long t = i / WP;
    const int hp = (int)(t % HP), n = (int)(t / HP);
    const int h = hp - 3, w = wp - 3;
    uint2 o = make_uint2(0, 0);
    if (h >= 0 && h < H && w >= 0 && w < W) {
      const size_t plane = (size_t)H * W;
      const float* src = x + (size_t)n * 3 * plane + (size_t)h * W + w;
      o.x = pack_bf16x2(src[0], src[plane]);
      o.y = pack_bf16x2(src[2 * plane], 0.f);
    }
    *reinterpret_cast<uint2*>(y + i * 4) = o;
  }
}

}  // namespace

// Forward convolution, NHWC bf16, square kernel k, stride, pad; optional BN-statistic partials.
// x: [N,IH,IW,*] with pixel stride ldx; w: [Cout][k*k][Cin] bf16; y: [N,OH,OW,*] with pixel stride ldy.
// stats (nullable): [ceil(M/BM)][2][Cout] fp32 partial sums where the number of partial rows is
// returned by yv1_conv2d_stats_rows().
extern "C" int yv1_conv2d_fwd_nhwc_bf16(const void* x, const void* w, void* y, int N, int IH, int IW, int ldx, int Cin,
                                        int Cout, int ldy, int k, int stride, int pad, float* stats,
                                        hipStream_t stream) {
  yv1_cfg_reset();
  if (!x || !w || !y || N <= 0 || k <= 0 || stride <= 0) return YV1_ERR_BAD_ARG;
  ConvArgs a;
  a.AS = nullptr; a.AM = nullptr; a.ldas = a.ldam = 0;
  a.escale = a.eshift = nullptr; a.ERES = nullptr; a.ldres = 0; a.erelu = 0;
  a.X = (const bf16_t*)x; a.W = (const bf16_t*)w; a.Y = (bf16_t*)y; a.stats = stats;
  a.N = N; a.IH = IH; a.IW = IW; a.ldx = ldx;
  a.P = (IH + 2 * pad - k) / stride + 1; a.Q = (IW + 2 * pad - k) / stride + 1;
  a.Cin = Cin; a.Cout = Cout; a.R = k; a.S = k;
  a.ah = stride; a.bh = 1; a.ch = -pad; a.aw = stride; a.bw = 1; a.cw = -pad; a.log2d = 0;
  a.OH = a.P; a.OW = a.Q; a.ldy = ldy; a.os = 1; a.accumulate = 0;
  a.oh0 = a.ow0 = 0; a.wr0 = a.ws0 = 0; a.wrs = a.wss = 1; a.WS = k; a.Kw = k * k * Cin;
  a.M = N * a.P * a.Q;
  return dispatch(a, stream);
}

// Inference form of the forward convolution: conv + folded eval-mode BatchNorm (scale/shift per output channel) +
// residual add + ReLU in one launch -- OriginResNet.py:87-107 with the network in eval() mode:
//   t = bf16(acc * scale[c] + shift[c]);  y = bf16(relu?(t + residual))        (no residual: ReLU before the rounding)
extern "C" int yv1_conv2d_fwd_bn_act_nhwc_bf16(const void* x, const void* w, void* y, int N, int IH, int IW, int ldx, int Cin,
                                               int Cout, int ldy, int k, int stride, int pad, const float* scale,
                                               const float* shift, const void* residual, int ldres, int relu,
                                               hipStream_t stream) {
  yv1_cfg_reset();
  if (!x || !w || !y || !scale || !shift || N <= 0 || k <= 0 || stride <= 0) return YV1_ERR_BAD_ARG;
  if (residual && ldres % 8) return YV1_ERR_UNSUPPORTED;
  ConvArgs a;
  a.AS = nullptr; a.AM = nullptr; a.ldas = a.ldam = 0;
  a.X = (const bf16_t*)x; a.W = (const bf16_t*)w; a.Y = (bf16_t*)y; a.stats = nullptr;
  a.N = N; a.IH = IH; a.IW = IW; a.ldx = ldx;
  a.P = (IH + 2 * pad - k) / stride + 1; a.Q = (IW + 2 * pad - k) / stride + 1;
  a.Cin = Cin; a.Cout = Cout; a.R = k; a.S = k;
  a.ah = stride; a.bh = 1; a.ch = -pad; a.aw = stride; a.bw = 1; a.cw = -pad; a.log2d = 0;
  a.OH = a.P; a.OW = a.Q; a.ldy = ldy; a.os = 1; a.accumulate = 0;
  a.oh0 = a.ow0 = 0; a.wr0 = a.ws0 = 0; a.wrs = a.wss = 1; a.WS = k; a.Kw = k * k * Cin;
  a.M = N * a.P * a.Q;
  a.escale = scale; a.eshift = shift; a.ERES = (const bf16_t*)residual; a.ldres = ldres; a.erelu = relu;
  return dispatch(a, stream);
}

// Stem: 7x7 stride-2 pad-3 convolution of the packed NHWC4 image (yv1_pack_input_nhwc4).
// xp: [N][H+6][W+6][4] bf16; w: [Cout][7][32] bf16 (element s*4+c of row r; zero for c==3 and s==7).
static int stem_forward(const void* xp, const void* w, void* y, int N, int H, int W, int Cout, int ldy, float* stats,
                        const float* scale, const float* shift, int relu, hipStream_t stream) {
  yv1_cfg_reset();
  if (!xp || !w || !y || N <= 0 || (H & 1) || (W & 1)) return YV1_ERR_BAD_ARG;
  ConvArgs a;
  a.AS = nullptr; a.AM = nullptr; a.ldas = a.ldam = 0;
  a.escale = scale; a.eshift = shift; a.ERES = nullptr; a.ldres = 0; a.erelu = relu;
  a.X = (const bf16_t*)xp; a.W = (const bf16_t*)w; a.Y = (bf16_t*)y; a.stats = stats;
  a.N = N; a.IH = H + 6; a.IW = W + 6; a.ldx = 4;       // "pixel" = 4 elements; one tap row = 32 contiguous elements
  a.P = H / 2; a.Q = W / 2;
  a.Cin = 32; a.Cout = Cout; a.R = 7; a.S = 1;
  a.ah = 2; a.bh = 1; a.ch = 0; a.aw = 2; a.bw = 0; a.cw = 0; a.log2d = 0;
  a.OH = a.P; a.OW = a.Q; a.ldy = ldy; a.os = 1; a.accumulate = 0;
  a.oh0 = a.ow0 = 0; a.wr0 = a.ws0 = 0; a.wrs = 1; a.wss = 0; a.WS = 1; a.Kw = 7 * 32;
  a.M = N * a.P * a.Q;
  a.dbg = 0;
  if (Cout % 64) return YV1_ERR_UNSUPPORTED;
  // the last tap row of the last pixel reads up to element ((IH-1)*IW + 2*(Q-1))*4 + 31 < IH*IW*4
  return launch<128, 64, 32, 2, 2>(a, stream);
}

extern "C" int yv1_conv2d_stem_fwd_bf16(const void* xp, const void* w, void* y, int N, int H, int W, int Cout, int ldy,
                                        float* stats, hipStream_t stream) {
  return stem_forward(xp, w, y, N, H, W, Cout, ldy, stats, nullptr, nullptr, 0, stream);
}

// inference form of the stem: + folded eval-mode BatchNorm + ReLU (OriginResNet.py:174-176 in eval mode)
extern "C" int yv1_conv2d_stem_fwd_bn_act_bf16(const void* xp, const void* w, void* y, int N, int H, int W, int Cout, int ldy,
                                               const float* scale, const float* shift, int relu, hipStream_t stream) {
  if (!scale || !shift) return YV1_ERR_BAD_ARG;
  return stem_forward(xp, w, y, N, H, W, Cout, ldy, nullptr, scale, shift, relu, stream);
}

// Data gradient.  dy: [N,OH,OW,*] (pixel stride lddy, Cout channels); wt: [Cin][k*k][Cout] bf16 (the
// transposed copy); dx: [N,IH,IW,*] (pixel stride lddx, Cin channels).  accumulate != 0 adds into dx.
// 1x1 strided convolutions touch only the pixels (h*stride, w*stride) -- with accumulate == 0 the other
// pixels of dx are NOT written (the caller zero-fills or, as the residual shortcut does, accumulates
// into a dx that the main path has already written).
extern "C" int yv1_conv2d_dgrad_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW, int lddx,
                                          int Cin, int Cout, int lddy, int k, int stride, int pad, int accumulate,
                                          hipStream_t stream) {
  yv1_cfg_reset();
  if (!dy || !wt || !dx || N <= 0 || k <= 0 || stride <= 0) return YV1_ERR_BAD_ARG;
  if (stride != 1 && stride != 2) return YV1_ERR_UNSUPPORTED;
  const int OH = (IH + 2 * pad - k) / stride + 1, OW = (IW + 2 * pad - k) / stride + 1;
  ConvArgs a;
  a.AS = nullptr; a.AM = nullptr; a.ldas = a.ldam = 0;
  a.escale = a.eshift = nullptr; a.ERES = nullptr; a.ldres = 0; a.erelu = 0;
  a.X = (const bf16_t*)dy; a.W = (const bf16_t*)wt; a.Y = (bf16_t*)dx; a.stats = nullptr;
  a.N = N; a.IH = OH; a.IW = OW; a.ldx = lddy;
  a.Cin = Cout; a.Cout = Cin; a.R = k; a.S = k;
  a.OH = IH; a.OW = IW; a.ldy = lddx; a.accumulate = accumulate;
  a.oh0 = a.ow0 = 0; a.wr0 = a.ws0 = 0; a.wrs = a.wss = 1; a.WS = k; a.Kw = k * k * Cout;
  if (k == 1 && pad == 0) {
    // GEMM over the dy pixels, scattered to (h*stride, w*stride)
    a.P = OH; a.Q = OW; a.os = stride;
    a.ah = 1; a.bh = 0; a.ch = 0; a.aw = 1; a.bw = 0; a.cw = 0; a.log2d = 0;
    a.M = N * a.P * a.Q;
    return dispatch(a, stream);
  }
  if (stride == 2 && k == 3 && pad == 1 && !(IH & 1) && !(IW & 1)) {
    // Output-parity decomposition: dx[2p+ph, 2q+pw] only receives the filter taps r with (ph + 1 - r) even.
    //   ph == 0: r = 1,      dy row p              ph == 1: r = 0 -> dy row p+1 ; r = 2 -> dy row p
    // Four launches, each a dense small convolution over the (IH/2 x IW/2) pixels of its class: no zero taps
    // are multiplied (the single generic launch below spends 3/4 of its MFMA work on them).
    a.P = IH / 2; a.Q = IW / 2; a.os = 2; a.log2d = 0;
    a.M = N * a.P * a.Q;
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw) {
        a.oh0 = ph; a.ow0 = pw;
        a.R = ph ? 2 : 1; a.S = pw ? 2 : 1;
        a.ah = 1; a.bh = ph ? -1 : 0; a.ch = ph ? 1 : 0; a.wr0 = ph ? 0 : 1; a.wrs = ph ? 2 : 0;
        a.aw = 1; a.bw = pw ? -1 : 0; a.cw = pw ? 1 : 0; a.ws0 = pw ? 0 : 1; a.wss = pw ? 2 : 0;
        const int rc = dispatch(a, stream);
        if (rc) return rc;
      }
    return YV1_OK;
  }
  // generic: GEMM over the dx pixels; tap (r,s) reads dy at ((h + pad - r)/stride, (w + pad - s)/stride)
  a.P = IH; a.Q = IW; a.os = 1;
  a.ah = 1; a.bh = -1; a.ch = pad; a.aw = 1; a.bw = -1; a.cw = pad; a.log2d = (stride == 2) ? 1 : 0;
  a.M = N * a.P * a.Q;
  return dispatch(a, stream);
}

// 1x1 stride-1 data gradient of the first convolution of an identity-shortcut Bottleneck, with the shortcut's own
// gradient folded into the epilogue:  dx = dgrad(dy, wt) + (relu_mask ? g : 0)  (OriginResNet.py:104-105 backward:
// out = relu(bn3(..) + x) hands x the gradient g where the output was positive).  g: [N,IH,IW,*] bf16 (pixel stride
// ldg), relu_mask: the 1-bit-per-element mask yv1_bn_apply wrote for that block output ([pixels][ldmask] bytes).
extern "C" int yv1_conv2d_dgrad_add_masked_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW,
                                                     int lddx, int Cin, int Cout, int lddy, const void* g, int ldg,
                                                     const void* relu_mask, int ldmask, hipStream_t stream) {
  yv1_cfg_reset();
  if (!dy || !wt || !dx || !g || !relu_mask || N <= 0) return YV1_ERR_BAD_ARG;
  if (ldg % 8 || Cin % 8) return YV1_ERR_UNSUPPORTED;
  ConvArgs a;
  a.escale = a.eshift = nullptr; a.ERES = nullptr; a.ldres = 0; a.erelu = 0;
  a.X = (const bf16_t*)dy; a.W = (const bf16_t*)wt; a.Y = (bf16_t*)dx; a.stats = nullptr;
  a.N = N; a.IH = IH; a.IW = IW; a.ldx = lddy;
  a.Cin = Cout; a.Cout = Cin; a.R = 1; a.S = 1;
  a.OH = IH; a.OW = IW; a.ldy = lddx; a.accumulate = 0;
  a.oh0 = a.ow0 = 0; a.wr0 = a.ws0 = 0; a.wrs = a.wss = 1; a.WS = 1; a.Kw = Cout;
  a.P = IH; a.Q = IW; a.os = 1;
  a.ah = 1; a.bh = 0; a.ch = 0; a.aw = 1; a.bw = 0; a.cw = 0; a.log2d = 0;
  a.M = N * a.P * a.Q;
  a.AS = (const bf16_t*)g; a.ldas = ldg; a.AM = (const unsigned char*)relu_mask; a.ldam = ldmask;
  return dispatch(a, stream);
}

// ---- "bn3 as algebra" (DESIGN.md section 7): the two convolution launches of it ------------------------------------
// conv1's data gradient of an identity-shortcut Bottleneck as yv1_conv2d_dgrad_add_masked_nhwc_bf16, with two additions for
// the block BELOW (whose output gradient dx is):  out_mask (nullable) -- that block's 1-bit ReLU mask: closed channels are
// stored as zero, i.e. dx leaves this launch as the MASKED gradient gm its BatchNorm-3 backward starts from;  gsum
// (nullable) -- [yv1_conv2d_dgrad_gsum_rows(M, Cin, Cout)][Cin] per-tile column sums of gm (its sum is bn3's dbeta).
// relu_mask may be NULL: g is then added as it is (it already is a masked gradient).
extern "C" int yv1_conv2d_dgrad_add_masked_out_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW,
                                                         int lddx, int Cin, int Cout, int lddy, const void* g, int ldg,
                                                         const void* relu_mask, int ldmask, const void* out_mask,
                                                         int ldom, float* gsum, hipStream_t stream) {
  yv1_cfg_reset();
  if (!dy || !wt || !dx || !g || N <= 0) return YV1_ERR_BAD_ARG;
  if (ldg % 8 || Cin % 8) return YV1_ERR_UNSUPPORTED;
  ConvArgs a;
  a.escale = a.eshift = nullptr; a.ERES = nullptr; a.ldres = 0; a.erelu = 0;
  a.X = (const bf16_t*)dy; a.W = (const bf16_t*)wt; a.Y = (bf16_t*)dx; a.stats = nullptr;
  a.N = N; a.IH = IH; a.IW = IW; a.ldx = lddy;
  a.Cin = Cout; a.Cout = Cin; a.R = 1; a.S = 1;
  a.OH = IH; a.OW = IW; a.ldy = lddx; a.accumulate = 0;
  a.oh0 = a.ow0 = 0; a.wr0 = a.ws0 = 0; a.wrs = a.wss = 1; a.WS = 1; a.Kw = Cout;
  a.P = IH; a.Q = IW; a.os = 1;
  a.ah = 1; a.bh = 0; a.ch = 0; a.aw = 1; a.bw = 0; a.cw = 0; a.log2d = 0;
  a.M = N * a.P * a.Q;
  a.AS = (const bf16_t*)g; a.ldas = ldg; a.AM = (const unsigned char*)relu_mask; a.ldam = ldmask;
  a.OM = (const unsigned char*)out_mask; a.ldom = ldom; a.gsum = gsum;
  return dispatch(a, stream);
}

// yv1_conv2d_dgrad_nhwc_bf16 for 1x1 / pad-0 convolutions (any stride 1 | 2) with the same two additions: the pair of data
// gradients a PROJECTION Bottleneck ends with (conv1's, then the strided downsample convolution's scatter-accumulate) hands
// the block below its output gradient masked, and each launch reports the column sums of what IT added (gsum rows:
// yv1_conv2d_dgrad_gsum_rows(N*OH*OW, Cin, Cout) with OH, OW the dy grid).
extern "C" int yv1_conv2d_dgrad_out_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW, int lddx, int Cin,
                                              int Cout, int lddy, int stride, int accumulate, const void* out_mask, int ldom,
                                              float* gsum, hipStream_t stream) {
  yv1_cfg_reset();
  if (!dy || !wt || !dx || N <= 0) return YV1_ERR_BAD_ARG;
  if (stride != 1 && stride != 2) return YV1_ERR_UNSUPPORTED;
  const int OH = (IH - 1) / stride + 1, OW = (IW - 1) / stride + 1;
  ConvArgs a;
  a.AS = nullptr; a.AM = nullptr; a.ldas = a.ldam = 0;
  a.escale = a.eshift = nullptr; a.ERES = nullptr; a.ldres = 0; a.erelu = 0;
  a.X = (const bf16_t*)dy; a.W = (const bf16_t*)wt; a.Y = (bf16_t*)dx; a.stats = nullptr;
  a.N = N; a.IH = OH; a.IW = OW; a.ldx = lddy;
  a.Cin = Cout; a.Cout = Cin; a.R = 1; a.S = 1;
  a.OH = IH; a.OW = IW; a.ldy = lddx; a.accumulate = accumulate;
  a.oh0 = a.ow0 = 0; a.wr0 = a.ws0 = 0; a.wrs = a.wss = 1; a.WS = 1; a.Kw = Cout;
  a.P = OH; a.Q = OW; a.os = stride;
  a.ah = 1; a.bh = 0; a.ch = 0; a.aw = 1; a.bw = 0; a.cw = 0; a.log2d = 0;
  a.M = N * a.P * a.Q;
  a.OM = (const unsigned char*)out_mask; a.ldom = ldom; a.gsum = gsum;
  return dispatch(a, stream);
}

// Deferred BatchNorm backward (OriginDenseNet.py:22-27 norm1 -> relu1 -> conv1, :50-52 the transition's norm -> relu -> conv;
// DESIGN.md section 7).  The 1x1 stride-1 pad-0 data gradient of a convolution whose input is relu(bn(x)), with the
// reduction-FREE part of that BatchNorm's backward in the epilogue:
//     d = bf16(dgrad(dy, wt)) where scale*x + shift > 0, else 0          (the gradient at the BatchNorm output)
//     dx = (accumulate ? dx : 0) + scale * d                              (bf16, one rounding)
//     part[tile][0][c] = sum_pixels d,  part[tile][1][c] = sum_pixels d * (x - mean[c])
// What the BatchNorm backward subtracts from it -- scale * (mean(d) + xhat * mean(d * xhat)), an affine function of x per
// channel -- comes out of yv1_bn_bwd_finalize_deferred as coefficients (A, B) and is applied LATE: by the next launch of this
// entry into the same buffer (pend_a / pend_b: dx -= A + B * x in the same epilogue pass -- in a dense block every layer's
// data gradient covers all channels below it, so the buffer never holds more than one uncorrected term) and, for the channels
// no later launch touches, by yv1_bn_deferred_fix before they are consumed.  The stand-alone reduce and apply passes and the
// stored d tensor disappear.
// x: the BatchNorm input [N,IH,IW,*] (pixel stride ldx), scale/shift/mean: its forward coefficients [Cin]; Cout (the
// convolution's output channels = GEMM K) must be a multiple of 64.  part: [yv1_conv2d_dgrad_bn_deferred_rows()][2][Cin].
// wt_rows: rows of wt ([wt_rows][Cout] bf16) that may be read, >= Cin and ZERO beyond Cin -- padded to a multiple of 128 the
// Cin = 64j + 32 layers run 64- / 128-wide column tiles whose last tile reaches past Cin.
extern "C" int yv1_conv2d_dgrad_bn_deferred_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW, int lddx,
                                                      int Cin, int Cout, int lddy, const void* x, int ldx, const float* scale,
                                                      const float* shift, const float* mean, int accumulate, float* part,
                                                      int wt_rows, const float* pend_a, const float* pend_b,
                                                      hipStream_t stream) {
  yv1_cfg_reset();
  if (!dy || !wt || !dx || !x || !scale || !shift || !mean || !part || N <= 0 || wt_rows < Cin) return YV1_ERR_BAD_ARG;
  if ((pend_a == nullptr) != (pend_b == nullptr) || (pend_a && !accumulate)) return YV1_ERR_BAD_ARG;
  if (Cout % 64 || Cin % 32 || ldx % 8 || lddx % 8) return YV1_ERR_UNSUPPORTED;
  ConvArgs a;
  a.X = (const bf16_t*)dy; a.W = (const bf16_t*)wt; a.Y = (bf16_t*)dx;
  a.N = N; a.IH = IH; a.IW = IW; a.ldx = lddy;
  a.Cin = Cout; a.Cout = Cin; a.R = 1; a.S = 1;
  a.OH = IH; a.OW = IW; a.ldy = lddx; a.accumulate = accumulate ? 1 : 0;
  a.oh0 = a.ow0 = 0; a.wr0 = a.ws0 = 0; a.wrs = a.wss = 1; a.WS = 1; a.Kw = Cout;
  a.P = IH; a.Q = IW; a.os = 1;
  a.ah = 1; a.bh = 0; a.ch = 0; a.aw = 1; a.bw = 0; a.cw = 0; a.log2d = 0;
  a.M = N * IH * IW;
  a.DBX = (const bf16_t*)x; a.lddbx = ldx; a.dbscale = scale; a.dbshift = shift; a.dbmean = mean; a.dbpart = part;
  a.db_wt_rows = wt_rows; a.dbpa = pend_a; a.dbpb = pend_b;
  return dispatch(a, stream);
}

// partial rows of yv1_conv2d_dgrad_bn_deferred_nhwc_bf16's sums: one per pixel tile of the kernel it dispatches
extern "C" int yv1_conv2d_dgrad_bn_deferred_rows(int M, int Cin, int Cout, int wt_rows) {
  const ConvPlan p = plan_deferred(M, Cin, Cout, wt_rows);
  return (M + p.bm - 1) / p.bm;
}

// Data gradient of a STRIDE-1 convolution (k x k, pad) whose input was relu(bn(y)), with the reduction pass of that
// BatchNorm's backward folded into the epilogue -- the mirror image of the forward's statistics epilogue:
//     dx = bf16(dgrad(dy, wt)) where scale*y + shift > 0, else 0        (the MASKED gradient d, stored)
//     part[tile][0][c] = sum d,   part[tile][1][c] = invstd[c] * sum d * (y - mean[c])  = sum d * xhat
// part has the layout yv1_bn_bwd_reduce writes, so yv1_bn_bwd_finalize + yv1_bn_bwd_apply (mask_mode 0: d is masked
// already) complete the BatchNorm backward; yv1_bn_bwd_reduce's pass over (dx, y) does not run.  Returns
// YV1_ERR_UNSUPPORTED for shapes whose plain data gradient runs a kernel without this epilogue (stride 2, the 256-wide tiles):
// yv1_conv2d_dgrad_bn_sums_rows() returns 0 for them and the caller keeps yv1_conv2d_dgrad_nhwc_bf16 + the reduce pass.
extern "C" int yv1_conv2d_dgrad_bn_sums_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW, int lddx,
                                                  int Cin, int Cout, int lddy, int k, int pad, const void* y, int ldy,
                                                  const float* scale, const float* shift, const float* mean,
                                                  const float* invstd, float* part, hipStream_t stream) {
  yv1_cfg_reset();
  if (!dy || !wt || !dx || !y || !scale || !shift || !mean || !invstd || !part || N <= 0 || k <= 0) return YV1_ERR_BAD_ARG;
  if (Cin % 32 || Cout % 32 || ldy % 8 || lddx % 8 || 2 * pad != k - 1) return YV1_ERR_UNSUPPORTED;
  ConvArgs a;
  a.X = (const bf16_t*)dy; a.W = (const bf16_t*)wt; a.Y = (bf16_t*)dx;
  a.N = N; a.IH = IH; a.IW = IW; a.ldx = lddy;                  // same-size output: the dy grid is the dx grid
  a.Cin = Cout; a.Cout = Cin; a.R = k; a.S = k;
  a.OH = IH; a.OW = IW; a.ldy = lddx; a.accumulate = 0;
  a.oh0 = a.ow0 = 0; a.wr0 = a.ws0 = 0; a.wrs = a.wss = 1; a.WS = k; a.Kw = k * k * Cout;
  a.P = IH; a.Q = IW; a.os = 1;
  if (k == 1) { a.ah = 1; a.bh = 0; a.ch = 0; a.aw = 1; a.bw = 0; a.cw = 0; }
  else { a.ah = 1; a.bh = -1; a.ch = pad; a.aw = 1; a.bw = -1; a.cw = pad; }
  a.log2d = 0;
  a.M = N * IH * IW;
  a.DBX = (const bf16_t*)y; a.lddbx = ldy; a.dbscale = scale; a.dbshift = shift; a.dbmean = mean; a.dbpart = part;
  a.db_unit = 1; a.dbis = invstd; a.db_wt_rows = Cin;
  return dispatch(a, stream);
}

// partial rows of yv1_conv2d_dgrad_bn_sums_nhwc_bf16 (one per pixel tile), 0 when the shape has no such kernel
extern "C" int yv1_conv2d_dgrad_bn_sums_rows(int M, int Cin, int Cout, int k, int pad) {
  if (Cin % 32 || Cout % 32 || 2 * pad != k - 1) return 0;
  const ConvPlan p = plan_db(M, Cin, Cout, k * k, k == 3 && pad == 1, Cin);
  if (p.kind != 1 && p.kind != 3) return 0;
  if (p.kind == 1 && !((p.bm == 128 && (p.bn == 128 || p.bn == 64 || p.bn == 32)) || (p.bm == 64 && p.bn == 64))) return 0;
  return (M + p.bm - 1) / p.bm;
}

// partial rows of gsum for a 1x1 data gradient with M pixels, Cin (= GEMM columns) and Cout (= GEMM K): one per pixel tile
extern "C" int yv1_conv2d_dgrad_gsum_rows(int M, int Cin, int Cout) {
  const ConvPlan p = plan_conv(M, Cin, Cout, 1);
  return (M + p.bm - 1) / p.bm;
}

// dx[M][Cdx] = [g | z] * wcat^T + bias:  the 1x1 stride-1 data gradient of conv3 with BatchNorm-3's backward folded into
// its operands.  g: [N,H,W,*] bf16 (C1 channels, pixel stride ldg), z: [N,H,W,*] bf16 (C2 channels, pixel stride ldz),
// wcat: bf16 [Cdx][C1 + C2] (K contiguous), bias: fp32 [Cdx] (one: fp32 [Cdx] of ones -- the epilogue's scale slot).
extern "C" int yv1_conv2d_dgrad_cat_bias_nhwc_bf16(const void* g, int ldg, int C1, const void* z, int ldz, int C2,
                                                   const void* wcat, const float* one, const float* bias, void* dx, int lddx,
                                                   int Cdx, int N, int H, int W, int stride, int accumulate,
                                                   const void* out_mask, int ldom, float* gsum, hipStream_t stream) {
  // g, z: [N,H,W,*] (the GEMM pixels); dx: [N, H*stride', W*stride', *] with pixel (h, w) of the GEMM stored at
  // (h*stride, w*stride) -- stride 2: the projection shortcut's scatter (dx spatial size = IH x IW given by H*stride... the
  // caller passes the GEMM grid H x W and dx is (H-1)*stride+1 .. rounded up to even: IH = H*stride for stride 2)
  yv1_cfg_reset();
  if (!g || !z || !wcat || !one || !bias || !dx || N <= 0) return YV1_ERR_BAD_ARG;
  if (C1 % 64 || C2 % 64 || ldg % 8 || ldz % 8 || lddx % 8 || Cdx % 32 || (stride != 1 && stride != 2)) return YV1_ERR_UNSUPPORTED;
  ConvArgs a;
  a.AS = nullptr; a.AM = nullptr; a.ldas = a.ldam = 0;
  a.ERES = nullptr; a.ldres = 0; a.erelu = 0;
  a.escale = one; a.eshift = bias;                       // out = bf16(acc * 1 + bias[n])
  a.X = (const bf16_t*)g; a.W = (const bf16_t*)wcat; a.Y = (bf16_t*)dx; a.stats = nullptr;
  a.N = N; a.IH = H; a.IW = W; a.ldx = ldg;
  a.Cin = C1 + C2; a.Cout = Cdx; a.R = 1; a.S = 1;
  a.OH = H * stride; a.OW = W * stride; a.ldy = lddx; a.accumulate = accumulate;
  a.oh0 = a.ow0 = 0; a.wr0 = a.ws0 = 0; a.wrs = a.wss = 1; a.WS = 1; a.Kw = C1 + C2;
  a.P = H; a.Q = W; a.os = stride;
  a.ah = 1; a.bh = 0; a.ch = 0; a.aw = 1; a.bw = 0; a.cw = 0; a.log2d = 0;
  a.M = N * H * W;
  a.X2 = (const bf16_t*)z; a.ldx2 = ldz;
  a.OM = (const unsigned char*)out_mask; a.ldom = ldom; a.gsum = gsum;      // as yv1_conv2d_dgrad_out_nhwc_bf16
  {
    const ConvPlan p = plan_conv(a.M, a.Cout, a.Cin, 1);
    if (p.kind == 0 || C1 % p.bk) return YV1_ERR_UNSUPPORTED;
    a.cb_split = C1 / p.bk;
  }
  return dispatch(a, stream);
}

extern "C" int yv1_conv2d_stats_rows(int M, int Cout, int Cin, int k, int stride, int pad) {
  // number of partial rows the forward kernel writes for this shape (same plan as dispatch()): one per pixel tile, or
  // one per workgroup slot for the persistent kernel
  const ConvPlan p = plan_conv(M, Cout, Cin, k * k, k == 3 && stride == 1 && pad == 1);
  const int MT = (M + p.bm - 1) / p.bm;
  if (p.kind == 2) {
    int per_cu = 0;                                            // statistics are written by the PLAIN (training) form only
#define YV1_RING_CASE(BM_, BN_, BK_, WM_, WN_, NST_)                                                             \
    if (p.bm == BM_ && p.bn == BN_ && p.bk == BK_ && p.nst == NST_) per_cu = ps_resident_per_cu<BM_, BN_, BK_, WM_, WN_, NST_, true>();
    YV1_RING_CASES
#undef YV1_RING_CASE
    if (!per_cu) return MT;
    return ps_slots(MT, Cout / p.bn, per_cu);
  }
  return MT;
}

extern "C" int yv1_pack_input_nhwc4(const float* x_nchw, void* y, int N, int H, int W, hipStream_t stream) {
  if (!x_nchw || !y || N <= 0) return YV1_ERR_BAD_ARG;
  const long long total = (long long)N * (H + 6) * (W + 6);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_pack_input, dim3(blocks), dim3(256), 0, stream, x_nchw, (bf16_t*)y, N, H, W);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}
