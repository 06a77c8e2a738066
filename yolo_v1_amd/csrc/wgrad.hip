// Convolution weight gradient on MFMA for gfx950 (MI355X / CDNA4).
//
// Replaces the cuDNN wgrad behind every nn.Conv2d of the reference backbones
// (backbones/OriginResNet.py:21-29,:121,:159-163; backbones/OriginDenseNet.py:24-29,:52-53) that
// `loss.backward()` runs at train.py:171.
//
// GEMM view:  dW[k][tap][c] = sum_m dY[m][k] * Xg[m,tap][c]
//   rows  = output channel k, cols = input channel c (one tap per workgroup), reduction = pixels m.
// Both operands live in HBM pixel-major (NHWC: channels contiguous), but MFMA wants the reduction
// index (pixels) along each lane's 8-element fragment.  The tiles are staged pixel-major into LDS
// with full 16-B channel runs (coalesced) and read back TRANSPOSED with ds_read_b64_tr_b16, the
// CDNA4 hardware transpose read: no shuffles, no scalar LDS reads.  LDS rows are padded by 64 B so
// the four pixel rows a transposed read touches fall on disjoint bank quarters.
// The pixel reduction is split over workgroups (split-K); each split writes an fp32 slab with plain
// stores and a second tiny kernel sums the slabs in a fixed order -- bitwise reproducible, no float
// atomics (which run at ~1/5 of the store rate on this chip).
// The result layout [Cout][taps][Cin] fp32 is exactly the channels_last storage of the OIHW
// parameter's .grad, so the optimizer consumes it without a layout pass.
// Kernels: k_wgrad (generic, one tap per workgroup, register-staged), k_wgrad_dma (1x1 layers: LDS-DMA ring,
// unpadded rows with XOR-swizzled 32-B segments), k_wgrad3x3 (3x3 stride 1: all nine taps per workgroup over an X
// halo tile), k_wgrad_stem (7x7/2 stem: all seven filter rows per workgroup), k_reduce_slabs.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short short4_;

struct WgradArgs {
  const bf16_t* X;    // [N,IH,IW,*] pixel stride ldx
  const bf16_t* DY;   // [N,P,Q,*]   pixel stride lddy
  float* OUT;         // slabs [splitK][Cout][taps][Cin] (or the final tensor when splitK == 1)
  int N, IH, IW, ldx, P, Q, lddy;
  int Cin, Cout, R, S;
  int ah, bh, ch, aw, bw, cw;   // ih = p*ah + r*bh + ch   (forward tap map)
  int M;                        // N*P*Q
  int splitK, steps_per_split;  // K-steps (of 32 pixels) per split
  int CT, KT;                   // cin tiles, cout tiles
  int dbg;                      // tuning only: bit0 skip in-loop loads, bit1 skip MFMA block
};


__device__ __forceinline__ bf16x4 lds_read_tr16(const unsigned char* p) {
  typedef __attribute__((address_space(3))) short4_ lds_s4;
  short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p));
  return __builtin_bit_cast(bf16x4, v);
}

template <int BMC, int BNC, int WM, int WN, int KP>
__global__ void __launch_bounds__(256, 2) k_wgrad(WgradArgs a) {
  // LDS row pitches (bytes): the 4 pixel rows of a transposed read must land on disjoint 16-dword bank quarters,
  // i.e. pitch = 16 or 48 (mod 64) dwords: 64-B rows need no padding, wider rows get +64 B
  constexpr int PA = BMC == 32 ? 64 : BMC * 2 + 64, PB = BNC == 32 ? 64 : BNC * 2 + 64;
  constexpr int A_BYTES = KP * PA, B_BYTES = KP * PB;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int ACH = BMC / 8, BCH = BNC / 8;             // 16-B chunks per pixel row
  constexpr int A_PASSES = (KP * ACH + 255) / 256, B_PASSES = (KP * BCH + 255) / 256;
  constexpr int TM = BMC / WM / 32, TN = BNC / WN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;

  // block -> (split, cout tile, cin tile, tap), taps innermost, through an XCD-aware (bijective) remap: the
  // workgroups that share an XCD (blockIdx % 8) get CONSECUTIVE logical ids, so the taps / channel tiles that
  // re-read one dY / X pixel range run next to each other on one L2 instead of being dealt over all eight
  // (PMC: 2.2x the algorithmic operand bytes were fetched without it)
  // Measured per layer: a clear win on the large feature maps (56x56, 112x112: up to 1.6x), a small loss on
  // the 28x28-and-below layers whose operands sit in the Infinity Cache anyway -> only used for M >= 100k pixels.
  int b = blockIdx.x;
  if (a.M >= 100000) {
    const int nwg = gridDim.x, o = blockIdx.x;
    const int xcd = o & 7, q = nwg >> 3, r8 = nwg & 7;
    b = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (o >> 3);
  }
  const int taps = a.R * a.S;
  const int tap = b % taps; b /= taps;
  const int ct = b % a.CT; b /= a.CT;
  const int kt_ = b % a.KT; b /= a.KT;
  const int split = b;
  const int r = tap / a.S, s = tap - r * a.S;
  const int k0 = kt_ * BMC, c0 = ct * BNC;

  // steps_per_split counts KP-pixel steps of THIS instantiation (the host plans with the same KP)
  const int step0 = split * a.steps_per_split;
  const int total_steps = (a.M + KP - 1) / KP;
  const int nsteps = min(a.steps_per_split, total_steps - step0);

  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 ra[A_PASSES], rb[B_PASSES];
  const int PQ = a.P * a.Q;
  constexpr int A_ROWSTEP = 256 / ACH, B_ROWSTEP = 256 / BCH;   // rows covered per pass
  const int a_row = tid / ACH, a_cc = tid % ACH;
  const int b_row = tid / BCH, b_cc = tid % BCH;
  const bool a_cok = k0 + a_cc * 8 < a.Cout;
  const bool b_cok = c0 + b_cc * 8 < a.Cin;           // the last Cin tile may be partial (Cin % BNC != 0)
  // x-operand pixel state per pass, advanced incrementally by KP pixels per K-step (no divisions in the loop)
  int xn[B_PASSES], xp[B_PASSES], xq[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int m = step0 * KP + b_row + i * B_ROWSTEP;
    const int n = m / PQ, rem = m - n * PQ;
    xn[i] = n; xp[i] = rem / a.Q; xq[i] = rem - xp[i] * a.Q;
  }
  int ld_m = step0 * KP;               // first pixel of the K-step the loader fetches next

#define YV1_WG_LOAD()                                                                                            \
  {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
      const int row = a_row + i * A_ROWSTEP;                                                                     \
      const int m = ld_m + row;                                                                                  \
      const bool ok = row < KP && m < a.M && a_cok;                                                              \
      const u32x4 v = *reinterpret_cast<const u32x4*>(a.DY + (ok ? (size_t)m * a.lddy + k0 + a_cc * 8 : (size_t)0)); \
      const u32x4 z = {0u, 0u, 0u, 0u};                                                                          \
      ra[i] = ok ? v : z;                                                                                        \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i) {                                                       \
      const int row = b_row + i * B_ROWSTEP;                                                                     \
      const int ih = xp[i] * a.ah + r * a.bh + a.ch, iw = xq[i] * a.aw + s * a.bw + a.cw;                        \
      const bool ok = row < KP && ld_m + row < a.M && ih >= 0 && ih < a.IH && iw >= 0 && iw < a.IW && b_cok;     \
      const size_t off = ok ? ((size_t)(xn[i] * a.IH + ih) * a.IW + iw) * a.ldx + c0 + b_cc * 8 : (size_t)0;     \
      const u32x4 v = *reinterpret_cast<const u32x4*>(a.X + off);                                                \
      const u32x4 z = {0u, 0u, 0u, 0u};                                                                          \
      rb[i] = ok ? v : z;                                                                                        \
      xq[i] += KP;                                                                                               \
      while (xq[i] >= a.Q) { xq[i] -= a.Q; ++xp[i]; }                                                            \
      while (xp[i] >= a.P) { xp[i] -= a.P; ++xn[i]; }                                                            \
    }                                                                                                            \
    ld_m += KP;                                                                                                  \
  }
#define YV1_WG_STORE(BUF_)                                                                                       \
  {                                                                                                              \
    unsigned char* sa_ = smem + (BUF_) * STAGE;                                                                  \
    unsigned char* sb_ = sa_ + A_BYTES;                                                                          \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
      const int row = a_row + i * A_ROWSTEP;                                                                     \
      if (row < KP) *reinterpret_cast<u32x4*>(sa_ + row * PA + a_cc * 16) = ra[i];                               \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i) {                                                       \
      const int row = b_row + i * B_ROWSTEP;                                                                     \
      if (row < KP) *reinterpret_cast<u32x4*>(sb_ + row * PB + b_cc * 16) = rb[i];                               \
    }                                                                                                            \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // transposed-read lane geometry (ds_read_b64_tr_b16): within each group of 16 lanes, lane 4q+p
  // supplies the address of pixel row q, channels 4p..4p+3; lane i receives channel i of the 4 rows.
  const int g = lane >> 4, li = lane & 15;
  const int tq = li >> 2, tp = li & 3;
  const int h = g >> 1;                               // which 8-pixel half of the 16-pixel MFMA K
  const int chan_off = 16 * (g & 1) + 4 * tp;         // channel offset inside a 32-channel block

  if (nsteps > 0) {
    YV1_WG_LOAD();
    YV1_WG_STORE(0);
  }
  __syncthreads();

  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    const bool do_ld = (st + 1 < nsteps) && !(a.dbg & 1);
    if (do_ld) YV1_WG_LOAD();
    const unsigned char* sa = smem + cur * STAGE;
    const unsigned char* sb = sa + A_BYTES;
    if (!(a.dbg & 2))
#pragma unroll
    for (int ks = 0; ks < KP / 16; ++ks) {
      const int prow = ks * 16 + 8 * h + tq;
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int col = wm * (BMC / WM) + i * 32 + chan_off;
        const bf16x4 lo = lds_read_tr16(sa + prow * PA + col * 2);
        const bf16x4 hi = lds_read_tr16(sa + (prow + 4) * PA + col * 2);
        fa[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * (BNC / WN) + j * 32 + chan_off;
        const bf16x4 lo = lds_read_tr16(sb + prow * PB + col * 2);
        const bf16x4 hi = lds_read_tr16(sb + (prow + 4) * PB + col * 2);
        fb[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (do_ld) YV1_WG_STORE(cur ^ 1);
    __syncthreads();
  }

  // epilogue: D[row = cout][col = cin]; col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int l31 = lane & 31, lh = lane >> 5;
  const size_t Ktot = (size_t)taps * a.Cin;
  float* out = a.OUT + (size_t)split * a.Cout * Ktot;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = c0 + wn * (BNC / WN) + j * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = k0 + wm * (BMC / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (k < a.Cout && c < a.Cin) out[(size_t)k * Ktot + (size_t)tap * a.Cin + c] = acc[i][j][e];
      }
    }
}

// ---- LDS-DMA ring form of k_wgrad (square 64x64 / 128x128 tiles, full channel tiles) --------------------------
// Same GEMM, same transposed fragment reads, but the pixel-major tiles travel global -> LDS directly
// (global_load_lds_dwordx4) into a ring of three stages with the next two K-steps in flight across a raw barrier,
// like conv.hip's k_conv_dma.  A DMA instruction writes 1 KB of consecutive LDS, so the rows cannot be padded.  A
// 32-lane group of a transposed read touches 4 pixel rows x 2 adjacent 32-byte segments; unpadded rows alias on the
// 256-byte bank row, so the byte offset inside a row is XOR-ed with (row & 3) << SEGSH -- 64-byte granularity for the
// 256-byte rows of the 128-wide tile (8 distinct segments), 32-byte for the 128-byte rows of the 64-wide tile (whose
// odd rows already sit on the other half of the bank row) -- on the source side (which 16-B chunk a lane fetches) and
// in the fragment-read address.  PMC: SQ_LDS_BANK_CONFLICT 0 % of SQ_LDS_IDX_ACTIVE (50 % with a 32-byte XOR on the
// 256-byte rows).
__device__ __attribute__((aligned(16))) unsigned g_wg_zero_page[4];

template <int NW>
__device__ __forceinline__ void wg_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((NW & 15) | ((NW >> 4) << 14) | (7 << 4) | (15 << 8));
}

template <int BT, bool LIN>
__global__ void __launch_bounds__(256, 2) k_wgrad_dma(WgradArgs a) {
  constexpr int KP = 32, NST = 3;
  constexpr int ROWB = BT * 2;                         // bytes per pixel row of a tile
  constexpr int CHR = BT / 8;                          // 16-B chunks per row
  constexpr int RPP = 64 / CHR;                        // rows per 1 KB piece
  constexpr int PIECES = KP / RPP;                     // pieces per operand tile
  constexpr int PPW = PIECES / 4;                      // pieces per wave and operand
  constexpr int LPS = 2 * PPW;
  constexpr int A_BYTES = KP * ROWB, STAGE = 2 * A_BYTES;
  constexpr int TM = BT / 64, TN = BT / 64;
  constexpr int SEGSH = BT == 128 ? 6 : 5;             // log2 of the XOR granularity in bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: DMA destinations stay in SGPRs
  const int wm = wid >> 1, wn = wid & 1;

  int b = blockIdx.x;
  if (a.M >= 100000) {
    const int nwg = gridDim.x, o = blockIdx.x;
    const int xcd = o & 7, q = nwg >> 3, r8 = nwg & 7;
    b = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (o >> 3);
  }
  const int taps = a.R * a.S;
  const int tap = b % taps; b /= taps;
  const int ct = b % a.CT; b /= a.CT;
  const int kt_ = b % a.KT; b /= a.KT;
  const int split = b;
  const int r = tap / a.S, s = tap - r * a.S;
  const int k0 = kt_ * BT, c0 = ct * BT;
  const int step0 = split * a.steps_per_split;
  const int total_steps = (a.M + KP - 1) / KP;
  const int nsteps = min(a.steps_per_split, total_steps - step0);

  // DMA lane geometry: row (lane / CHR) of the piece, physical chunk lane % CHR holds logical chunk lchunk
  const int lrow = lane / CHR;
  const int lchunk = (lane % CHR) ^ ((lrow & 3) << (SEGSH - 4));
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_wg_zero_page);
  const int PQ = a.P * a.Q;
  int xn[PPW], xp[PPW], xq[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int m = step0 * KP + (wid + 4 * i) * RPP + lrow;
    const int n = m / PQ, rem = m - n * PQ;
    xn[i] = n; xp[i] = rem / a.Q; xq[i] = rem - xp[i] * a.Q;
  }
  // Loader state.  dY rows are consecutive pixels: one pointer per piece, advanced by KP pixels per K-step.  The X rows
  // are too when the convolution has stride 1 (every 1x1 layer but the four projection shortcuts): `lin`; otherwise
  // the (n, p, q) state of the generic kernel is kept.  Rows past M read the zero page.
  constexpr bool lin = LIN;                            // host: a.ah == 1 && a.aw == 1 && a.ch == 0 && a.cw == 0
  int mrow[PPW];
  const bf16_t* pa[PPW];
  const bf16_t* pb[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    mrow[i] = step0 * KP + (wid + 4 * i) * RPP + lrow;
    pa[i] = a.DY + ((size_t)mrow[i] * a.lddy + k0 + lchunk * 8);
    pb[i] = a.X + ((size_t)mrow[i] * a.ldx + c0 + lchunk * 8);
  }
  const bool cok = c0 + lchunk * 8 < a.Cin;            // partial last Cin tile (DenseNet bottlenecks): those lanes read zeros
  const size_t stepa = (size_t)KP * a.lddy, stepb = (size_t)KP * a.ldx;
#define YV1_WGD_ISSUE(STG_)                                                                                      \
  {                                                                                                              \
    unsigned char* sa_ = smem + (STG_) * STAGE;                                                                  \
    unsigned char* sb_ = sa_ + A_BYTES;                                                                          \
    _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                            \
      const int piece = wid + 4 * i;                                                                             \
      const bool inm = mrow[i] < a.M;                                                                            \
      const bf16_t* srca = inm ? pa[i] : zsrc;                                                                   \
      __builtin_amdgcn_global_load_lds((glb_void*)srca, (lds_void*)(sa_ + piece * 1024), 16, 0, 0);              \
      const bf16_t* srcb;                                                                                        \
      if (lin) {                                                                                                 \
        srcb = (inm && cok) ? pb[i] : zsrc;                                                                      \
      } else {                                                                                                   \
        const int ih = xp[i] * a.ah + r * a.bh + a.ch, iw = xq[i] * a.aw + s * a.bw + a.cw;                      \
        const bool ok = inm && cok && ih >= 0 && ih < a.IH && iw >= 0 && iw < a.IW;                              \
        srcb = ok ? a.X + (((size_t)(xn[i] * a.IH + ih) * a.IW + iw) * a.ldx + c0 + lchunk * 8) : zsrc;          \
        xq[i] += KP;                                                                                             \
        while (xq[i] >= a.Q) { xq[i] -= a.Q; ++xp[i]; }                                                          \
        while (xp[i] >= a.P) { xp[i] -= a.P; ++xn[i]; }                                                          \
      }                                                                                                          \
      __builtin_amdgcn_global_load_lds((glb_void*)srcb, (lds_void*)(sb_ + piece * 1024), 16, 0, 0);              \
      pa[i] += stepa; pb[i] += stepb; mrow[i] += KP;                                                             \
    }                                                                                                            \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int g = lane >> 4, li = lane & 15;
  const int tq = li >> 2, tp = li & 3;
  const int h = g >> 1;
  const int chan_off = 16 * (g & 1) + 4 * tp;
  const int xr = tq << SEGSH;                          // (pixel row & 3) << SEGSH: the row's XOR, in bytes
  // per-lane byte offsets of the transposed fragment reads inside a stage, computed once (the stage base is an
  // immediate in the unrolled loop)
  constexpr int KS = KP / 16;
  int fa_off[TM][KS], fb_off[TN][KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int prow = ks * 16 + 8 * h + tq;             // prow & 3 == tq, also for prow + 4
#pragma unroll
    for (int i = 0; i < TM; ++i) fa_off[i][ks] = prow * ROWB + (((wm * (BT / 2) + i * 32 + chan_off) * 2) ^ xr);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb_off[j][ks] = A_BYTES + prow * ROWB + (((wn * (BT / 2) + j * 32 + chan_off) * 2) ^ xr);
  }
#define YV1_WGD_MFMA(BASE_)                                                                                      \
  _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                            \
    bf16x8 fa[TM], fb[TN];                                                                                       \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                             \
      const bf16x4 lo = lds_read_tr16((BASE_) + fa_off[i][ks]);                                                  \
      const bf16x4 hi = lds_read_tr16((BASE_) + fa_off[i][ks] + 4 * ROWB);                                       \
      fa[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                                           \
    }                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                             \
      const bf16x4 lo = lds_read_tr16((BASE_) + fb_off[j][ks]);                                                  \
      const bf16x4 hi = lds_read_tr16((BASE_) + fb_off[j][ks] + 4 * ROWB);                                       \
      fb[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                                           \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                               \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                             \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);                   \
  }

  if (nsteps > 0) YV1_WGD_ISSUE(0);
  if (nsteps > 1) YV1_WGD_ISSUE(1);
  int st = 0;
  // steady state, unrolled over the three-stage ring (constant stage indices): one K-step stays in flight, every step
  // issues the one two ahead
  for (; st + NST <= nsteps - 2; st += NST) {
#pragma unroll
    for (int c = 0; c < NST; ++c) {
      // fragment reads of the previous step complete (and are not scheduled below) the barrier: the DMA after it refills
      // that stage.  s_barrier alone is no memory fence to the compiler (see conv.hip).
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wg_wait_vmcnt<LPS>();
      __builtin_amdgcn_s_barrier();
      YV1_WGD_ISSUE((c + 2) % NST);
      YV1_WGD_MFMA(smem + c * STAGE);
    }
  }
  int cur = 0, nxt = 2;                                // st is a multiple of NST here
  for (; st < nsteps; ++st) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (st + 1 < nsteps) wg_wait_vmcnt<LPS>(); else wg_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (st + 2 < nsteps) YV1_WGD_ISSUE(nxt);
    YV1_WGD_MFMA(smem + cur * STAGE);
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }
#undef YV1_WGD_MFMA
#undef YV1_WGD_ISSUE

  const int l31 = lane & 31, lh = lane >> 5;
  const size_t Ktot = (size_t)taps * a.Cin;
  float* out = a.OUT + (size_t)split * a.Cout * Ktot;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = c0 + wn * (BT / 2) + j * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = k0 + wm * (BT / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (c < a.Cin) out[(size_t)k * Ktot + (size_t)tap * a.Cin + c] = acc[i][j][e];
      }
    }
}

// out[i] = sum_s slabs[s][i].  256 threads = 64 float4 columns x 4 split lanes: a workgroup reads 1 KB contiguous runs of
// every slab (the first version read 256-byte runs: 2.3 TB/s on slabs that mostly sit in L2 / Infinity Cache), a lane
// sums every 4th slab with four independent loads in flight, the 4 partial sums are combined through LDS in a fixed
// order (bitwise reproducible).
__global__ void __launch_bounds__(256) k_reduce_slabs(const float* __restrict__ slabs, float* __restrict__ out, long long n,
                                                      int splitK) {
  __shared__ float4 red[4][64];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long long n4 = n >> 2;
  const long long i = (long long)blockIdx.x * 64 + col;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (i < n4) {
    const float4* base = reinterpret_cast<const float4*>(slabs) + i;
    int k = sl;
    for (; k + 12 < splitK; k += 16) {
      const float4 v0 = base[(size_t)k * n4], v1 = base[(size_t)(k + 4) * n4];
      const float4 v2 = base[(size_t)(k + 8) * n4], v3 = base[(size_t)(k + 12) * n4];
      s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
      s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
      s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
      s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
    }
    for (; k < splitK; k += 4) {
      const float4 v = base[(size_t)k * n4];
      s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
    }
  }
  float4 s = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                         (s0.w + s1.w) + (s2.w + s3.w));
  red[sl][col] = s;
  __syncthreads();
  if (sl == 0 && i < n4) {
    const float4 a1 = red[1][col], a2 = red[2][col], a3 = red[3][col];
    float4 t;
    t.x = (s.x + a1.x) + (a2.x + a3.x); t.y = (s.y + a1.y) + (a2.y + a3.y);
    t.z = (s.z + a1.z) + (a2.z + a3.z); t.w = (s.w + a1.w) + (a2.w + a3.w);
    reinterpret_cast<float4*>(out)[i] = t;
  }
}

// ---- stem (7x7/2, Cin packed to 4) weight gradient: ALL 7 filter rows in one workgroup ---------------------
// The generic kernel gives every filter row its own workgroups, so the 411 MB dY (N=64) is re-read 7 times and
// half of each 128-row cout tile is padding.  Here a workgroup owns the whole [64 cout][7 rows][32] result for a
// pixel range: dY is staged once per K-step and used against the 7 image rows (wave w: cout half w&1, filter rows
// {0..3} or {4..6}).  Split-K over pixels into fp32 slabs as above.
struct StemWgradArgs {
  const bf16_t* XP;   // [N][H+6][W+6][4]
  const bf16_t* DY;   // [N][P][Q][*], pixel stride lddy, 64 channels
  float* OUT;         // slabs [split][64][7][32]
  int IHp, IWp, P, Q, lddy, M, steps_per_split;
};

__global__ void __launch_bounds__(256, 2) k_wgrad_stem(StemWgradArgs a) {
  constexpr int KPS = 32;
  constexpr int PA = 64 * 2 + 64, PB = 32 * 2;                 // LDS pitches (bytes): 48 / 16 dwords = conflict-free tr reads
  constexpr int A_BYTES = KPS * PA, B_BYTES = 7 * KPS * PB, STAGE = A_BYTES + B_BYTES;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int chalf = wid & 1, tgrp = wid >> 1;
  const int t0 = tgrp ? 4 : 0, nt = tgrp ? 3 : 4;
  const int split = blockIdx.x;
  const int step0 = split * a.steps_per_split;
  const int total_steps = (a.M + KPS - 1) / KPS;
  const int nsteps = min(a.steps_per_split, total_steps - step0);
  const int PQ = a.P * a.Q;

  // loaders: dY 32 rows x 8 chunks = 256 chunks (1 per thread); image 7 x 32 rows x 4 chunks = 896 (4 passes)
  const int a_row = tid >> 3, a_cc = tid & 7;
  const int b_cc = tid & 3, b_row = (tid >> 2) & 31, b_t0 = tid >> 7;          // tap = b_t0 + 2*pass
  int xn, xp, xq;                                                              // pixel of row b_row
  {
    const int m = step0 * KPS + b_row;
    xn = m / PQ; const int rem = m - xn * PQ; xp = rem / a.Q; xq = rem - xp * a.Q;
  }
  int ld_m = step0 * KPS;
  u32x4 ra, rb[4];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#define YV1_STEM_LOAD()                                                                                          \
  {                                                                                                              \
    const int ma = ld_m + a_row;                                                                                 \
    const bool oka = ma < a.M;                                                                                   \
    const u32x4 va = *reinterpret_cast<const u32x4*>(a.DY + (oka ? (size_t)ma * a.lddy + a_cc * 8 : (size_t)0)); \
    ra = oka ? va : zero4;                                                                                       \
    const bool okb = ld_m + b_row < a.M;                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
      const int tap = b_t0 + 2 * i;                                                                              \
      const bool ok = okb && tap < 7;                                                                            \
      const size_t off = ok ? ((size_t)(xn * a.IHp + 2 * xp + tap) * a.IWp + 2 * xq) * 4 + b_cc * 8 : (size_t)0; \
      const u32x4 v = *reinterpret_cast<const u32x4*>(a.XP + off);                                               \
      rb[i] = ok ? v : zero4;                                                                                    \
    }                                                                                                            \
    xq += KPS;                                                                                                   \
    while (xq >= a.Q) { xq -= a.Q; ++xp; }                                                                       \
    while (xp >= a.P) { xp -= a.P; ++xn; }                                                                       \
    ld_m += KPS;                                                                                                 \
  }
#define YV1_STEM_STORE(BUF_)                                                                                     \
  {                                                                                                              \
    unsigned char* sa_ = smem + (BUF_) * STAGE;                                                                  \
    unsigned char* sb_ = sa_ + A_BYTES;                                                                          \
    *reinterpret_cast<u32x4*>(sa_ + a_row * PA + a_cc * 16) = ra;                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
      const int tap = b_t0 + 2 * i;                                                                              \
      if (tap < 7) *reinterpret_cast<u32x4*>(sb_ + (tap * KPS + b_row) * PB + b_cc * 16) = rb[i];                \
    }                                                                                                            \
  }
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3, h = g >> 1;
  const int chan_off = 16 * (g & 1) + 4 * tp;
  if (nsteps > 0) { YV1_STEM_LOAD(); YV1_STEM_STORE(0); }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    if (st + 1 < nsteps) YV1_STEM_LOAD();
    const unsigned char* sa = smem + cur * STAGE;
    const unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < KPS / 16; ++ks) {
      const int prow = ks * 16 + 8 * h + tq;
      const int acol = chalf * 32 + chan_off;
      const bf16x4 alo = lds_read_tr16(sa + prow * PA + acol * 2);
      const bf16x4 ahi = lds_read_tr16(sa + (prow + 4) * PA + acol * 2);
      const bf16x8 fa = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < nt) {
          const unsigned char* tb = sb + (size_t)(t0 + t) * KPS * PB;
          const bf16x4 blo = lds_read_tr16(tb + prow * PB + chan_off * 2);
          const bf16x4 bhi = lds_read_tr16(tb + (prow + 4) * PB + chan_off * 2);
          const bf16x8 fb = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
        }
      }
    }
    if (st + 1 < nsteps) YV1_STEM_STORE(cur ^ 1);
    __syncthreads();
  }
  const int l31 = lane & 31, lh = lane >> 5;
  float* out = a.OUT + (size_t)split * 64 * 224;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < nt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = chalf * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        out[(size_t)k * 224 + (t0 + t) * 32 + l31] = acc[t][e];
      }
    }
  }
}


// ---- 3x3 stride-1 pad-1 weight gradient: ALL 9 taps in one workgroup ----------------------------------------
// The generic kernel gives every tap its own workgroups: dY and X are re-read (L2 -> LDS) nine times and each K-step
// stages 8 KB for 8 MFMAs.  Here a workgroup owns a [64 cout][9 taps][64 cin] result.  A K-step is KP consecutive
// pixels of ONE image row: dY is staged once ([KP][64]) next to the X halo ([3 rows][KP+2 pixels][64]) and the nine
// taps are nine row/column offsets into that halo -- 3.2x fewer operand bytes and 4.5x fewer LDS stores per MFMA.
// Each wave owns a 32x32 (cout x cin) quadrant and nine accumulators.  Split-K over image-row segments into fp32
// slabs, reduced in a fixed order by k_reduce_slabs like the generic path.
struct W3Args {
  const bf16_t* X;    // [N,H,W,*] pixel stride ldx
  const bf16_t* DY;   // [N,H,W,*] pixel stride lddy
  float* OUT;         // slabs [split][Cout][9][Cin]
  int N, H, W, ldx, lddy, Cin, Cout;
  int SPR;            // K-steps per image row = ceil(W / KP)
  int total_steps, steps_per_split, KT, CT;
};

// BMC x BNC = cout x cin channels of the result tile: 64x64 (2x2 waves) or 32x128 (1x4 waves, DenseNet's growth convs)
template <int KP, int BMC, int BNC>
__global__ void __launch_bounds__(256, 2) k_wgrad3x3(W3Args a) {
  static_assert((BMC / 32) * (BNC / 32) == 4, "four waves, one 32x32 quadrant and nine accumulators each");
  constexpr int HWD = KP + 2;                         // halo width (pixels)
  // row pitches = 16 or 48 (mod 64) dwords: conflict-free transposed reads (see k_wgrad)
  constexpr int PA = BMC == 32 ? 64 : BMC * 2 + 64, PB = BNC * 2 + 64;
  constexpr int ACH = BMC / 8, BCH = BNC / 8, WN = BNC / 32;
  constexpr int A_BYTES = KP * PA, B_BYTES = 3 * HWD * PB, STAGE = A_BYTES + B_BYTES;
  constexpr int HCHUNKS = 3 * HWD * BCH, B_PASSES = (HCHUNKS + 255) / 256;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;

  int b = blockIdx.x;
  {                                                   // XCD-aware (bijective) order: see k_wgrad
    const int nwg = gridDim.x, o = blockIdx.x;
    const int xcd = o & 7, q = nwg >> 3, r8 = nwg & 7;
    b = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (o >> 3);
  }
  const int ct = b % a.CT; b /= a.CT;
  const int kt_ = b % a.KT; b /= a.KT;
  const int split = b;
  const int k0 = kt_ * BMC, c0 = ct * BNC;
  const int step0 = split * a.steps_per_split;
  const int nsteps = min(a.steps_per_split, a.total_steps - step0);

  // loader state: (image, row, segment) of the K-step fetched next
  int ld_n, ld_h, ld_seg;
  {
    const int nh = step0 / a.SPR;
    ld_seg = step0 - nh * a.SPR;
    ld_n = nh / a.H;
    ld_h = nh - ld_n * a.H;
  }
  const int a_row = tid / ACH, a_cc = tid % ACH;
  int b_hr[B_PASSES], b_j[B_PASSES], b_cc[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int idx = tid + i * 256;
    const int hr = idx / (HWD * BCH), rem = idx - hr * (HWD * BCH);
    b_hr[i] = idx < HCHUNKS ? hr : -1;
    b_j[i] = rem / BCH;
    b_cc[i] = rem % BCH;
  }
  u32x4 ra, rb[B_PASSES];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#define YV1_W3_LOAD()                                                                                            \
  {                                                                                                              \
    const int w0 = ld_seg * KP;                                                                                  \
    const int rowbase = (ld_n * a.H + ld_h) * a.W;                                                               \
    {                                                                                                            \
      const bool ok = a_row < KP && w0 + a_row < a.W;                                                            \
      const u32x4 v = *reinterpret_cast<const u32x4*>(a.DY + (ok ? (size_t)(rowbase + w0 + a_row) * a.lddy + k0 + a_cc * 8 : (size_t)0)); \
      ra = ok ? v : zero4;                                                                                       \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i) {                                                       \
      const int ih = ld_h + b_hr[i] - 1, iw = w0 - 1 + b_j[i];                                                   \
      const bool ok = b_hr[i] >= 0 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;                                \
      const size_t off = ok ? (size_t)((ld_n * a.H + ih) * a.W + iw) * a.ldx + c0 + b_cc[i] * 8 : (size_t)0;     \
      const u32x4 v = *reinterpret_cast<const u32x4*>(a.X + off);                                                \
      rb[i] = ok ? v : zero4;                                                                                    \
    }                                                                                                            \
    if (++ld_seg == a.SPR) { ld_seg = 0; if (++ld_h == a.H) { ld_h = 0; ++ld_n; } }                              \
  }
#define YV1_W3_STORE(BUF_)                                                                                       \
  {                                                                                                              \
    unsigned char* sa_ = smem + (BUF_) * STAGE;                                                                  \
    unsigned char* sb_ = sa_ + A_BYTES;                                                                          \
    if (a_row < KP) *reinterpret_cast<u32x4*>(sa_ + a_row * PA + a_cc * 16) = ra;                                \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i) {                                                       \
      if (b_hr[i] >= 0) *reinterpret_cast<u32x4*>(sb_ + (b_hr[i] * HWD + b_j[i]) * PB + b_cc[i] * 16) = rb[i]; \
    }                                                                                                            \
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3, hh = g >> 1;
  const int chan_off = 16 * (g & 1) + 4 * tp;
  if (nsteps > 0) { YV1_W3_LOAD(); YV1_W3_STORE(0); }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    if (st + 1 < nsteps) YV1_W3_LOAD();
    const unsigned char* sa = smem + cur * STAGE;
    const unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < KP / 16; ++ks) {
      const int prow = ks * 16 + 8 * hh + tq;
      const int acol = (wm * 32 + chan_off) * 2, bcol = (wn * 32 + chan_off) * 2;
      const bf16x4 alo = lds_read_tr16(sa + prow * PA + acol);
      const bf16x4 ahi = lds_read_tr16(sa + (prow + 4) * PA + acol);
      const bf16x8 fa = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
          const unsigned char* tb = sb + (r * HWD + s_) * PB;       // tap (r,s): x[h + r - 1][w + s - 1]
          const bf16x4 blo = lds_read_tr16(tb + prow * PB + bcol);
          const bf16x4 bhi = lds_read_tr16(tb + (prow + 4) * PB + bcol);
          const bf16x8 fb = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[r * 3 + s_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[r * 3 + s_], 0, 0, 0);
        }
    }
    if (st + 1 < nsteps) YV1_W3_STORE(cur ^ 1);
    __syncthreads();
  }
#undef YV1_W3_LOAD
#undef YV1_W3_STORE

  const int l31 = lane & 31, lh = lane >> 5;
  const size_t Ktot = (size_t)9 * a.Cin;
  float* out = a.OUT + (size_t)split * a.Cout * Ktot;
  const int c = c0 + wn * 32 + l31;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = k0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      out[(size_t)k * Ktot + (size_t)t * a.Cin + c] = acc[t][e];
    }
}

struct Plan3 { bool use; int kp, spr, total_steps, steps, splitK, KT, CT, bmc; };

// shared: the launch overlaps with another stream's kernels (the backward's side stream) -- a narrower split-K then wins:
// fewer slabs to write and re-read, fewer workgroups taken from the other stream (in-step sweep, DESIGN.md section 7:
// 512/320 target workgroups beat 768/512 by 1.3 %); a launch that has the device to itself wants the wider split
// (serialized backward: 768/512 beat 512/320 by 2.9 % and 1024/768 by 5 %).
Plan3 make_plan3(int N, int H, int W, int Cin, int Cout, int k, int stride, int pad, bool shared) {
  Plan3 p;
  static int enabled = -1, want_shared = 0, want_alone = 0, thin_minw = 48;
  if (enabled < 0) {
    const char* tm = getenv("YV1_WGRAD3_THIN_MINW");  // tuning: narrowest map the thin (Cout 32) multi-tap form takes
    if (tm && atoi(tm) >= 8) thin_minw = atoi(tm);
    const char* e = getenv("YV1_WGRAD3");            // tuning: 0 disables the multi-tap kernel
    enabled = e ? atoi(e) : 1;
    const char* w = getenv("YV1_WGRAD3_BLOCKS");
    want_shared = w ? atoi(w) : 320;
    if (want_shared < 32) want_shared = 320;
    const char* wa = getenv("YV1_WGRAD3_BLOCKS_ALONE");
    want_alone = wa ? atoi(wa) : 512;
    if (want_alone < 32) want_alone = 512;
  }
  const int want_blocks = shared ? want_shared : want_alone;
  // measured (tools/bench_conv.py): 2.0x on 112x112, 1.3x on 56x56, 1.2x on 28x28; 16-pixel segments (14x14 maps) gain
  // nothing over the generic kernel, so rows shorter than 24 pixels stay there
  const bool square = Cin % 64 == 0 && Cout % 64 == 0 && W >= 24;
  // DenseNet growth convolutions (Cout 32): one 32x128 tile per 128 input channels.  The result is tiny
  // (32x9x128 fp32 = 147 KB per slab), so all the parallelism is split-K: worth it on the wide maps only
  const bool thin = Cout == 32 && Cin % 128 == 0 && W >= thin_minw;
  p.use = enabled && k == 3 && stride == 1 && pad == 1 && (square || thin);
  if (!p.use) return p;
  p.bmc = square ? 64 : 32;
  p.kp = 32;
  p.spr = (W + p.kp - 1) / p.kp;
  p.total_steps = N * H * p.spr;
  p.KT = Cout / p.bmc; p.CT = Cin / (square ? 64 : 128);
  int want = want_blocks / (p.KT * p.CT);
  if (want < 1) want = 1;
  int maxsplit = p.total_steps / (square ? 16 : 32);  // at least 16 (32) K-steps per split
  if (maxsplit < 1) maxsplit = 1;
  if (want > maxsplit) want = maxsplit;
  p.steps = (p.total_steps + want - 1) / want;
  p.splitK = (p.total_steps + p.steps - 1) / p.steps;
  return p;
}

template <int KP, int BMC, int BNC>
int launch3(W3Args& a, int nblocks, hipStream_t stream) {
  constexpr int STAGE = KP * (BMC == 32 ? 64 : BMC * 2 + 64) + 3 * (KP + 2) * (BNC * 2 + 64);
  if (2 * STAGE > 64 * 1024) YV1_SET_MAX_LDS((k_wgrad3x3<KP, BMC, BNC>), 2 * STAGE);
  yv1_cfg_note("k_wgrad3x3<%d,%d,%d>", KP, BMC, BNC);
  hipLaunchKernelGGL((k_wgrad3x3<KP, BMC, BNC>), dim3(nblocks), dim3(256), 2 * STAGE, stream, a);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

template <int BMC, int BNC, int WM, int WN, int KP>
int launch(WgradArgs& a, int nblocks, hipStream_t stream) {
  constexpr int STAGE = KP * (BMC == 32 ? 64 : BMC * 2 + 64) + KP * (BNC == 32 ? 64 : BNC * 2 + 64);
  if (2 * STAGE > 64 * 1024) YV1_SET_MAX_LDS((k_wgrad<BMC, BNC, WM, WN, KP>), 2 * STAGE);
  yv1_cfg_note("k_wgrad<%d,%d,%d,%d,%d>", BMC, BNC, WM, WN, KP);
  hipLaunchKernelGGL((k_wgrad<BMC, BNC, WM, WN, KP>), dim3(nblocks), dim3(256), 2 * STAGE, stream, a);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

struct Plan { int bmc, bnc, kp, KT, CT, splitK, steps; };

int wgrad_want_blocks(bool shared) {
  static int w = 0, wa = 0;
  if (!w) {
    const char* e = getenv("YV1_WGRAD_BLOCKS"); w = e ? atoi(e) : 512; if (w < 64) w = 512;
    const char* a = getenv("YV1_WGRAD_BLOCKS_ALONE"); wa = a ? atoi(a) : 768; if (wa < 64) wa = 768;
  }
  return shared ? w : wa;
}

int wgrad_kp() {
  static int kp = 0;
  if (!kp) { const char* e = getenv("YV1_WGRAD_KP"); kp = e ? atoi(e) : 32; if (kp != 32 && kp != 64 && kp != 128) kp = 32; }
  return kp;
}

Plan make_plan(int M, int Cin, int Cout, int taps, bool shared) {
  Plan p;
  // DenseNet's 1x1 bottlenecks (Cout 128, Cin = 64 + 32 i): a partial last 128-wide Cin tile (masked loads) beats
  // 32- or 64-wide tiles that re-read dY once per tile, as long as at most a quarter of the tile columns are padding
  const int cin_pad128 = (Cin + 127) / 128 * 128;
  if (Cout % 128 == 0 && (Cin % 128 == 0 || (taps == 1 && Cin % 32 == 0 && (cin_pad128 - Cin) * 4 <= cin_pad128))) {
    p.bmc = 128; p.bnc = 128;
  }
  else if (Cout % 64 == 0 && Cin % 64 == 0) { p.bmc = 64; p.bnc = 64; }
  else if (Cout <= 32 && Cin % 128 == 0) { p.bmc = 32; p.bnc = 128; }
  else { p.bmc = 128; p.bnc = 32; }
  p.KT = (Cout + p.bmc - 1) / p.bmc;
  p.CT = (Cin + p.bnc - 1) / p.bnc;
  p.kp = wgrad_kp();
  if (p.kp == 128 && p.bmc == 128 && p.bnc == 128) p.kp = 64;      // LDS budget
  const int tiles = p.KT * p.CT * taps;
  const int total_steps = (M + p.kp - 1) / p.kp;
  int want = (wgrad_want_blocks(shared) + tiles - 1) / tiles;   // 2-3 workgroups per CU; every split costs a slab written and read back
  int maxsplit = total_steps / (512 / p.kp);        // at least 512 pixels per split
  if (maxsplit < 1) maxsplit = 1;
  if (want > maxsplit) want = maxsplit;
  if (want < 1) want = 1;
  p.steps = (total_steps + want - 1) / want;
  p.splitK = (total_steps + p.steps - 1) / p.steps;
  return p;
}

template <int KP>
int run_plan_kp(const Plan& p, WgradArgs& a, int nblocks, hipStream_t stream) {
  if (p.bmc == 128 && p.bnc == 128) {
    if constexpr (KP <= 64) return launch<128, 128, 2, 2, KP>(a, nblocks, stream);
    else return YV1_ERR_UNSUPPORTED;
  }
  if (p.bmc == 64) return launch<64, 64, 2, 2, KP>(a, nblocks, stream);
  if (p.bmc == 32) return launch<32, 128, 1, 4, KP>(a, nblocks, stream);
  return launch<128, 32, 4, 1, KP>(a, nblocks, stream);
}

int run_plan(const Plan& p, WgradArgs& a, int nblocks, hipStream_t stream) {
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("YV1_WGRAD_DBG"); dbg = e ? atoi(e) : 0; }
    a.dbg = dbg;
  }
  {
    static int dma = -1;                     // YV1_WGRAD_DMA=0: register-staged loop
    if (dma < 0) { const char* e = getenv("YV1_WGRAD_DMA"); dma = e ? atoi(e) : 1; }
    // measured: 5-12 % faster on the 1x1 layers, slower on the per-tap workgroups of 3x3 layers -> 1x1 only
    if (dma && a.R * a.S == 1 && p.kp == 32 && p.bmc == p.bnc && (p.bmc == 128 || p.bmc == 64) && a.Cout % p.bmc == 0 &&
        (a.Cin % p.bnc == 0 || (p.bmc == 128 && a.Cin % 8 == 0)) && a.lddy % 8 == 0 && a.ldx % 8 == 0) {
      const bool lin = a.ah == 1 && a.aw == 1 && a.ch == 0 && a.cw == 0;     // X rows are consecutive pixels too
      yv1_cfg_note("k_wgrad_dma<%d,%s>", p.bmc, lin ? "true" : "false");
      if (p.bmc == 128) {
        YV1_SET_MAX_LDS((k_wgrad_dma<128, true>), 3 * 2 * 32 * 256);
        YV1_SET_MAX_LDS((k_wgrad_dma<128, false>), 3 * 2 * 32 * 256);
        if (lin) hipLaunchKernelGGL((k_wgrad_dma<128, true>), dim3(nblocks), dim3(256), 3 * 2 * 32 * 256, stream, a);
        else hipLaunchKernelGGL((k_wgrad_dma<128, false>), dim3(nblocks), dim3(256), 3 * 2 * 32 * 256, stream, a);
      } else {
        if (lin) hipLaunchKernelGGL((k_wgrad_dma<64, true>), dim3(nblocks), dim3(256), 3 * 2 * 32 * 128, stream, a);
        else hipLaunchKernelGGL((k_wgrad_dma<64, false>), dim3(nblocks), dim3(256), 3 * 2 * 32 * 128, stream, a);
      }
      YV1_LAUNCH_CHECK();
      return YV1_OK;
    }
  }
  if (p.kp == 32) return run_plan_kp<32>(p, a, nblocks, stream);
  if (p.kp == 128) return run_plan_kp<128>(p, a, nblocks, stream);
  return run_plan_kp<64>(p, a, nblocks, stream);
}

}  // namespace

extern "C" size_t yv1_conv2d_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int k) {
  // the 3x3 multi-tap kernel needs stride 1 / pad 1, which this size query does not see: reserve the larger of the two
  // plans for 3x3 shapes (the stride-1 plan is the larger one whenever it applies)
  size_t need3 = 0;
  if (k == 3) {
    for (int shared = 0; shared < 2; ++shared) {
      const Plan3 p3 = make_plan3(N, OH, OW, Cin, Cout, 3, 1, 1, shared != 0);
      const size_t n3 = p3.use ? (size_t)p3.splitK * Cout * 9 * Cin * sizeof(float) : 0;
      if (n3 > need3) need3 = n3;
    }
  }
  size_t need = 0;
  for (int shared = 0; shared < 2; ++shared) {             // one size serves both entry points
    const Plan p = make_plan(N * OH * OW, Cin, Cout, k * k, shared != 0);
    const size_t n = p.splitK > 1 ? (size_t)p.splitK * Cout * k * k * Cin * sizeof(float) : 0;
    if (n > need) need = n;
  }
  return need > need3 ? need : need3;
}

// dw[Cout][k*k][Cin] fp32 = sum over pixels of dy (x) x_tap.   x: [N,IH,IW,*] (pixel stride ldx),
// dy: [N,OH,OW,*] (pixel stride lddy).  workspace: yv1_conv2d_wgrad_workspace_bytes().
static int conv2d_wgrad(const void* x, const void* dy, float* dw, int N, int IH, int IW, int ldx, int Cin, int Cout, int lddy,
                        int k, int stride, int pad, void* workspace, size_t workspace_bytes, bool shared, hipStream_t stream) {
  yv1_cfg_reset();
  if (!x || !dy || !dw || N <= 0 || k <= 0 || stride <= 0) return YV1_ERR_BAD_ARG;
  if (Cin % 32 || ldx % 8 || lddy % 8 || Cout % 8) return YV1_ERR_UNSUPPORTED;
  WgradArgs a;
  a.X = (const bf16_t*)x; a.DY = (const bf16_t*)dy;
  a.N = N; a.IH = IH; a.IW = IW; a.ldx = ldx;
  a.P = (IH + 2 * pad - k) / stride + 1; a.Q = (IW + 2 * pad - k) / stride + 1; a.lddy = lddy;
  a.Cin = Cin; a.Cout = Cout; a.R = k; a.S = k;
  a.ah = stride; a.bh = 1; a.ch = -pad; a.aw = stride; a.bw = 1; a.cw = -pad;
  a.M = N * a.P * a.Q;
  const Plan3 p3 = make_plan3(N, IH, IW, Cin, Cout, k, stride, pad, shared);
  if (p3.use) {
    const size_t need3 = (size_t)p3.splitK * Cout * 9 * Cin * sizeof(float);
    if (need3 > workspace_bytes || !workspace) return YV1_ERR_WORKSPACE;
    W3Args w;
    w.X = (const bf16_t*)x; w.DY = (const bf16_t*)dy; w.OUT = (float*)workspace;
    w.N = N; w.H = IH; w.W = IW; w.ldx = ldx; w.lddy = lddy; w.Cin = Cin; w.Cout = Cout;
    w.SPR = p3.spr; w.total_steps = p3.total_steps; w.steps_per_split = p3.steps; w.KT = p3.KT; w.CT = p3.CT;
    const int nb = p3.splitK * p3.KT * p3.CT;
    const int rc3 = p3.bmc == 64 ? launch3<32, 64, 64>(w, nb, stream) : launch3<32, 32, 128>(w, nb, stream);
    if (rc3) return rc3;
    const long long n3 = (long long)Cout * 9 * Cin;
    yv1_cfg_note("k_reduce_slabs splitK=%d", p3.splitK);
    hipLaunchKernelGGL(k_reduce_slabs, dim3((int)((n3 / 4 + 63) / 64)), dim3(256), 0, stream, (const float*)workspace, dw,
                       n3, p3.splitK);
    YV1_LAUNCH_CHECK();
    return YV1_OK;
  }
  const Plan p = make_plan(a.M, Cin, Cout, k * k, shared);
  const size_t need = p.splitK > 1 ? (size_t)p.splitK * Cout * k * k * Cin * sizeof(float) : 0;
  if (need > workspace_bytes || (need && !workspace)) return YV1_ERR_WORKSPACE;
  a.splitK = p.splitK; a.steps_per_split = p.steps; a.CT = p.CT; a.KT = p.KT;
  a.OUT = p.splitK > 1 ? (float*)workspace : dw;
  const int nblocks = p.splitK * p.KT * p.CT * k * k;
  int rc = run_plan(p, a, nblocks, stream);
  if (rc) return rc;
  if (p.splitK > 1) {
    const long long n = (long long)Cout * k * k * Cin;
    const int blocks = (int)((n / 4 + 63) / 64);
    yv1_cfg_note("k_reduce_slabs splitK=%d", p.splitK);
    hipLaunchKernelGGL(k_reduce_slabs, dim3(blocks), dim3(256), 0, stream, (const float*)workspace, dw, n, p.splitK);
    YV1_LAUNCH_CHECK();
  }
  return YV1_OK;
}

extern "C" int yv1_conv2d_wgrad_nhwc_bf16(const void* x, const void* dy, float* dw, int N, int IH, int IW, int ldx,
                                          int Cin, int Cout, int lddy, int k, int stride, int pad, void* workspace,
                                          size_t workspace_bytes, hipStream_t stream) {
  return conv2d_wgrad(x, dy, dw, N, IH, IW, ldx, Cin, Cout, lddy, k, stride, pad, workspace, workspace_bytes, false, stream);
}

// The same gradient (bitwise different: another split-K width, i.e. another summation order) for a launch that overlaps with
// kernels of another stream -- the backward's weight-gradient stream beside the dgrad / BatchNorm chain: narrower split-K.
extern "C" int yv1_conv2d_wgrad_shared_nhwc_bf16(const void* x, const void* dy, float* dw, int N, int IH, int IW, int ldx,
                                                 int Cin, int Cout, int lddy, int k, int stride, int pad, void* workspace,
                                                 size_t workspace_bytes, hipStream_t stream) {
  return conv2d_wgrad(x, dy, dw, N, IH, IW, ldx, Cin, Cout, lddy, k, stride, pad, workspace, workspace_bytes, true, stream);
}

// Stem weight gradient: x is the packed NHWC4 image [N][H+6][W+6][4]; dw comes out as [Cout][7][32]
// (element s*4+c of filter row r), the layout yv1_conv2d_stem_fwd_bf16 consumes.
namespace {
int stem_plan(int M, int* steps) {
  const int total_steps = (M + 31) / 32;
  int splits = 1024;
  if (splits > total_steps / 8) splits = total_steps / 8 > 0 ? total_steps / 8 : 1;
  *steps = (total_steps + splits - 1) / splits;
  return (total_steps + *steps - 1) / *steps;
}
}  // namespace

extern "C" size_t yv1_conv2d_stem_wgrad_workspace_bytes(int N, int H, int W, int Cout) {
  int steps;
  const int splits = stem_plan(N * (H / 2) * (W / 2), &steps);
  return (size_t)splits * Cout * 7 * 32 * sizeof(float);
}

extern "C" int yv1_conv2d_stem_wgrad_bf16(const void* xp, const void* dy, float* dw, int N, int H, int W, int Cout,
                                          int lddy, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  yv1_cfg_reset();
  if (!xp || !dy || !dw || !workspace || N <= 0 || (H & 1) || (W & 1)) return YV1_ERR_BAD_ARG;
  if (Cout != 64 || lddy % 8) return YV1_ERR_UNSUPPORTED;
  StemWgradArgs a;
  a.XP = (const bf16_t*)xp; a.DY = (const bf16_t*)dy; a.OUT = (float*)workspace;
  a.IHp = H + 6; a.IWp = W + 6; a.P = H / 2; a.Q = W / 2; a.lddy = lddy;
  a.M = N * a.P * a.Q;
  const int splits = stem_plan(a.M, &a.steps_per_split);
  if ((size_t)splits * 64 * 224 * sizeof(float) > workspace_bytes) return YV1_ERR_WORKSPACE;
  constexpr int STAGE = 32 * (64 * 2 + 64) + 7 * 32 * (32 * 2);
  YV1_SET_MAX_LDS(k_wgrad_stem, 2 * STAGE);
  yv1_cfg_note("k_wgrad_stem");
  hipLaunchKernelGGL(k_wgrad_stem, dim3(splits), dim3(256), 2 * STAGE, stream, a);
  YV1_LAUNCH_CHECK();
  const long long n = 64ll * 7 * 32;
  hipLaunchKernelGGL(k_reduce_slabs, dim3((int)((n / 4 + 63) / 64)), dim3(256), 0, stream, (const float*)workspace, dw, n, splits);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}
