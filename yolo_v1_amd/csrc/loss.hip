// Fused YOLO-v1 grid loss, forward + backward, for gfx950.
//
// Replaces reference v1Loss.py:22-118 (YOLOLossV1.forward) and the autograd
// graph PyTorch builds behind it, including utils/utils.py:10-75 (IoU helpers)
// which the reference calls once per object cell from a Python loop.
//
// Semantics reproduced on purpose:
//   T4  "location" loss slices ROWS of the responsible-box list: the first two
//       responsible boxes in (n,i,j) order use plain squared error on x,y,w,h,
//       all later ones squared error of sqrt on x,y,w,h alike  (v1Loss.py:101)
//   T5  the IoU used as confidence target is not detached: its gradient flows
//       into the responsible box's x,y,w,h                     (v1Loss.py:72-78,:90)
//   T9  the total is divided by the constructor batch size     (v1Loss.py:105)
//
// Launch plan (one call = memset + 3 tiny launches on the caller's stream):
//   k_first2    one wave per 64 cells: ballot over "target conf0 == 1", merges
//               the two smallest object-cell indices into a 64-bit word (CAS).
//   k_main      one wave per 64 cells: stages pred/target rows through LDS with
//               coalesced loads, one lane per cell computes the four loss terms
//               and d(total)/d(pred) in registers, wavefront reductions give
//               one partial per block, gradient goes back out coalesced.
//   k_finalize  one wave: sums the per-block partials in a fixed order (bitwise
//               reproducible), writes the 4 component sums and the total.
// Algorithmic HBM bytes per image: 3 * S*S*(B*5+C) * 4 (read pred, read
// target, write grad) = 17.6 KB at S=7, 70.6 KB at S=14: latency-bound.
//
// Built with -ffp-contract=off so the IoU / argmax arithmetic is the same fp32
// op sequence as the reference's.
#include "common.h"

namespace {

constexpr int CELLS_PER_BLOCK = 64;
constexpr int MAX_B = 8;

__global__ void __launch_bounds__(64) k_first2(const float* __restrict__ target, int ncells, int D,
                                               unsigned long long* first2) {
  const int lane = threadIdx.x;
  const int cell = blockIdx.x * CELLS_PER_BLOCK + lane;
  const bool obj = cell < ncells && target[(size_t)cell * D] == 1.0f;   // v1Loss.py:28
  unsigned long long m = __ballot(obj);
  if (m == 0 || lane != 0) return;
  const unsigned base = blockIdx.x * CELLS_PER_BLOCK;
  unsigned a = base + (unsigned)__builtin_ctzll(m);
  m &= m - 1;
  unsigned b = m ? base + (unsigned)__builtin_ctzll(m) : 0xffffffffu;
  unsigned long long old = *first2, assumed;
  do {
    assumed = old;
    unsigned lo = (unsigned)assumed, hi = (unsigned)(assumed >> 32);
    // two smallest of {lo, hi, a, b}; lo<=hi, a<b, all distinct unless 0xffffffff
    unsigned n0 = min(lo, a);
    unsigned n1 = min(max(lo, a), min(hi, b));
    unsigned long long nw = ((unsigned long long)n1 << 32) | n0;
    if (nw == assumed) break;
    old = atomicCAS(first2, assumed, nw);
  } while (old != assumed);
}

struct LossArgs {
  const float* pred;
  long long ps0, ps1, ps2, ps3;   // pred strides in elements for [N,S,S,D]
  const float* target;            // contiguous [N,S,S,D]
  float* grad;                    // contiguous [N,S,S,D] or nullptr
  float* partials;                // [gridDim.x][4]  loc, hit, nohit, cls
  const unsigned long long* first2;
  int ncells, S, B, C;
  float l_coord, l_noobj, gscale;  // gscale = 1 / batch_size
  int pred_contig;
};

__global__ void __launch_bounds__(64) k_main(LossArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.B * 5 + a.C;
  float* sp = smem;                          // [64][D] pred, later grad
  float* st = smem + CELLS_PER_BLOCK * D;    // [64][D] target
  const int lane = threadIdx.x;
  const int base = blockIdx.x * CELLS_PER_BLOCK;
  const int cnt = min(CELLS_PER_BLOCK, a.ncells - base);
  const int tot = cnt * D;

  for (int i = lane; i < tot; i += 64) st[i] = a.target[(size_t)base * D + i];
  if (a.pred_contig) {
    for (int i = lane; i < tot; i += 64) sp[i] = a.pred[(size_t)base * D + i];
  } else {
    const int SS = a.S * a.S;
    for (int i = lane; i < tot; i += 64) {
      int c = base + i / D, d = i % D;
      int n = c / SS, r = c % SS;
      sp[i] = a.pred[n * a.ps0 + (r / a.S) * a.ps1 + (r % a.S) * a.ps2 + d * a.ps3];
    }
  }
  __syncthreads();

  float loc = 0.f, hit = 0.f, nohit = 0.f, cls = 0.f;
  if (lane < cnt) {
    float* p = sp + lane * D;
    const float* t = st + lane * D;
    const int B = a.B, C = a.C;
    const float g = a.gscale;
    const bool obj = t[0] == 1.0f;
    float conf[MAX_B];
#pragma unroll
    for (int b = 0; b < MAX_B; ++b) conf[b] = b < B ? p[b] : 0.f;

    if (!obj) {
      // every slot of a non-object cell is "not responsible": target IoU 0   (v1Loss.py:80,:91)
#pragma unroll
      for (int b = 0; b < MAX_B; ++b)
        if (b < B) { nohit += conf[b] * conf[b]; }
      for (int d = 0; d < D; ++d) p[d] = 0.f;
#pragma unroll
      for (int b = 0; b < MAX_B; ++b)
        if (b < B) p[b] = 2.f * conf[b] * a.l_noobj * g;
    } else {
      // ---- class term (v1Loss.py:33-41)
      for (int c = 0; c < C; ++c) {
        float d = p[5 * B + c] - t[5 * B + c];
        cls += d * d;
        p[5 * B + c] = 2.f * d * g;
      }
      // ---- IoU of every predicted box against gt slot 0 (v1Loss.py:66-74, utils.py:10-75)
      const float Sf = (float)a.S;
      const float gcx = t[B + 0] / Sf, gcy = t[B + 1] / Sf;
      const float ghw = 0.5f * t[B + 2], ghh = 0.5f * t[B + 3];
      const float gx1 = gcx - ghw, gy1 = gcy - ghh, gx2 = gcx + ghw, gy2 = gcy + ghh;
      const float a2 = (gx2 - gx1) * (gy2 - gy1);
      int r = 0;
      float best = 0.f;
      float bx1 = 0, by1 = 0, bx2 = 0, by2 = 0, biw = 0, bih = 0, bI = 0, bU = 1;
      for (int b = 0; b < B; ++b) {
        const float* pb = p + B + 4 * b;
        const float cx = pb[0] / Sf, cy = pb[1] / Sf;
        const float hw = 0.5f * pb[2], hh = 0.5f * pb[3];
        const float x1 = cx - hw, y1 = cy - hh, x2 = cx + hw, y2 = cy + hh;
        const float iw = fminf(x2, gx2) - fmaxf(x1, gx1);
        const float ih = fminf(y2, gy2) - fmaxf(y1, gy1);
        const float iwc = iw < 0.f ? 0.f : iw, ihc = ih < 0.f ? 0.f : ih;
        const float I = iwc * ihc;
        const float a1 = (x2 - x1) * (y2 - y1);
        const float U = (a1 + a2) - I;
        const float iou = I / U;
        if (b == 0 || iou > best) {   // first index wins ties (torch.max over dim)
          best = iou; r = b;
          bx1 = x1; by1 = y1; bx2 = x2; by2 = y2; biw = iw; bih = ih; bI = I; bU = U;
        }
      }
      // ---- confidence terms (v1Loss.py:90-91)
      const float dc = conf[r] - best;
      hit = dc * dc;
      for (int b = 0; b < B; ++b) {
        if (b == r) { p[b] = 2.f * dc * g; }
        else { nohit += conf[b] * conf[b]; p[b] = 2.f * conf[b] * a.l_noobj * g; }
      }
      // ---- location term with the row-slicing quirk (v1Loss.py:94-101)
      const unsigned f0 = (unsigned)(*a.first2), f1 = (unsigned)(*a.first2 >> 32);
      const unsigned me = (unsigned)(base + lane);
      const bool plain = (me == f0) || (me == f1);
      float gb[4];
      for (int k = 0; k < 4; ++k) {
        const float pv = p[B + 4 * r + k], tv = t[B + 4 * r + k];
        if (plain) {
          const float d = pv - tv;
          loc += d * d;
          gb[k] = 2.f * d * a.l_coord * g;
        } else {
          const float sp_ = sqrtf(pv), st_ = sqrtf(tv);
          const float d = sp_ - st_;
          loc += d * d;
          gb[k] = (d / sp_) * a.l_coord * g;     // 2*d * 1/(2*sqrt(p))
        }
      }
      // ---- gradient through the IoU target (T5): d hit / d iou = -2 (conf_r - iou)
      {
        const float giou = -2.f * dc * g;
        const float gI = giou * (1.f / bU + bI / (bU * bU));
        const float ga1 = -giou * bI / (bU * bU);
        const float iwc = biw < 0.f ? 0.f : biw, ihc = bih < 0.f ? 0.f : bih;
        const float giw = biw < 0.f ? 0.f : gI * ihc;
        const float gih = bih < 0.f ? 0.f : gI * iwc;
        // iw = min(x2,gx2) - max(x1,gx1); ties split the gradient evenly (torch maximum/minimum)
        const float wx1 = bx1 > gx1 ? 1.f : (bx1 == gx1 ? 0.5f : 0.f);
        const float wx2 = bx2 < gx2 ? 1.f : (bx2 == gx2 ? 0.5f : 0.f);
        const float wy1 = by1 > gy1 ? 1.f : (by1 == gy1 ? 0.5f : 0.f);
        const float wy2 = by2 < gy2 ? 1.f : (by2 == gy2 ? 0.5f : 0.f);
        const float w_ = bx2 - bx1, h_ = by2 - by1;
        const float gx1_ = -giw * wx1 - ga1 * h_;
        const float gx2_ = giw * wx2 + ga1 * h_;
        const float gy1_ = -gih * wy1 - ga1 * w_;
        const float gy2_ = gih * wy2 + ga1 * w_;
        gb[0] += (gx1_ + gx2_) / Sf;
        gb[1] += (gy1_ + gy2_) / Sf;
        gb[2] += 0.5f * (gx2_ - gx1_);
        gb[3] += 0.5f * (gy2_ - gy1_);
      }
      for (int b = 0; b < B; ++b)
        for (int k = 0; k < 4; ++k) p[B + 4 * b + k] = (b == r) ? gb[k] : 0.f;
    }
  }
  loc = wave_sum(loc); hit = wave_sum(hit); nohit = wave_sum(nohit); cls = wave_sum(cls);
  if (lane == 0) {
    float* o = a.partials + (size_t)blockIdx.x * 4;
    o[0] = loc; o[1] = hit; o[2] = nohit; o[3] = cls;
  }
  __syncthreads();
  if (a.grad)
    for (int i = lane; i < tot; i += 64) a.grad[(size_t)base * D + i] = sp[i];
}

__global__ void __launch_bounds__(64) k_finalize(const float* __restrict__ partials, int nblk, float l_coord,
                                                 float l_noobj, float batch, float* out_loss, float* out_comps) {
  const int lane = threadIdx.x;
  double s[4] = {0, 0, 0, 0};
  for (int i = lane; i < nblk; i += 64)
    for (int k = 0; k < 4; ++k) s[k] += (double)partials[(size_t)i * 4 + k];
  for (int k = 0; k < 4; ++k)
    for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
  if (lane == 0) {
    const float loc = (float)s[0], hit = (float)s[1], nohit = (float)s[2], cls = (float)s[3];
    out_comps[0] = loc; out_comps[1] = hit; out_comps[2] = nohit; out_comps[3] = cls;
    float t = l_coord * loc + hit;      // v1Loss.py:104
    t = t + l_noobj * nohit;
    t = t + cls;
    *out_loss = t / batch;              // v1Loss.py:105
  }
}

__global__ void k_scale_by_scalar(float* __restrict__ x, const float* __restrict__ s, long long n) {
  const float v = *s;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    x[i] *= v;
}

}  // namespace

extern "C" size_t yv1_loss_workspace_bytes(int N, int S) {
  const long long ncells = (long long)N * S * S;
  const long long nblk = (ncells + CELLS_PER_BLOCK - 1) / CELLS_PER_BLOCK;
  return 16 + (size_t)nblk * 4 * sizeof(float);
}

extern "C" int yv1_loss_fwd_bwd(const float* pred, long long ps0, long long ps1, long long ps2, long long ps3,
                                const float* target, int N, int S, int B, int C, float l_coord, float l_noobj,
                                float batch_size, float* out_loss, float* out_components, float* grad_pred,
                                void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!pred || !target || !out_loss || !out_components || !workspace) return YV1_ERR_BAD_ARG;
  if (N <= 0 || S <= 0 || B <= 0 || B > MAX_B || C < 0 || batch_size <= 0.f) return YV1_ERR_BAD_ARG;
  if (workspace_bytes < yv1_loss_workspace_bytes(N, S)) return YV1_ERR_WORKSPACE;
  const int D = B * 5 + C;
  const int ncells = N * S * S;
  const int nblk = (ncells + CELLS_PER_BLOCK - 1) / CELLS_PER_BLOCK;
  unsigned long long* first2 = (unsigned long long*)workspace;
  float* partials = (float*)((char*)workspace + 16);
  YV1_HIP(hipMemsetAsync(first2, 0xff, 16, stream));
  hipLaunchKernelGGL(k_first2, dim3(nblk), dim3(64), 0, stream, target, ncells, D, first2);
  YV1_LAUNCH_CHECK();
  LossArgs a;
  a.pred = pred; a.ps0 = ps0; a.ps1 = ps1; a.ps2 = ps2; a.ps3 = ps3;
  a.target = target; a.grad = grad_pred; a.partials = partials; a.first2 = first2;
  a.ncells = ncells; a.S = S; a.B = B; a.C = C;
  a.l_coord = l_coord; a.l_noobj = l_noobj; a.gscale = 1.0f / batch_size;
  a.pred_contig = (ps3 == 1 && ps2 == D && ps1 == (long long)S * D && ps0 == (long long)S * S * D) ? 1 : 0;
  const size_t lds = (size_t)2 * CELLS_PER_BLOCK * D * sizeof(float);
  hipLaunchKernelGGL(k_main, dim3(nblk), dim3(64), lds, stream, a);
  YV1_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, stream, partials, nblk, l_coord, l_noobj, batch_size,
                     out_loss, out_components);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

// x[i] *= *scalar  (chains an upstream autograd scalar into the saved gradient without a host sync)
extern "C" int yv1_scale_by_device_scalar(float* x, const float* scalar, long long n, hipStream_t stream) {
  if (!x || !scalar || n < 0) return YV1_ERR_BAD_ARG;
  if (n == 0) return YV1_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_scale_by_scalar, dim3(blocks), dim3(256), 0, stream, x, scalar, n);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}
