// Batched YOLO-v1 decoder + greedy class-agnostic NMS for gfx950, plus the
// stand-alone NMS and the pairwise-IoU / box-convert helpers.
//
// Replaces reference utils/utils.py:94-147 (decoder: a triple Python loop over
// S x S x B with ~10 tiny tensor ops and a host sync per slot) and :150-184
// (nms: a Python while-loop, one round per kept box), :10-75 (IoU helpers).
//
// One workgroup (256 threads = 4 wavefronts) per image, everything in LDS:
//   1. decode every slot (i,j,b): mask, box -> xyxy, score = conf * max class,
//      double-precision threshold compare exactly as the reference's
//      `float(score) > thresh`                                   (:108-132)
//   2. order-preserving compaction of the surviving slots (ballot + prefix)
//   3. rank sort by (score desc, index asc)  -- n <= S*S*B <= 392, O(n^2/256)
//   4. upper-triangular suppression bit-matrix: bit (r,c) set iff NOT
//      (IoU(r,c) <= thr), same fp32 op order as :170-180, so NaN suppresses
//   5. one wavefront walks the sorted list: lane w owns alive-word w
//   6. outputs in keep order: boxes, class, score, candidate index
// Zero candidates yield the reference's single all-zero box (:134-137).
// The reference's in-place write into `pred` is not reproduced.
// Algorithmic HBM bytes per image: S*S*(B*5+C)*4 in, <= S*S*B*36 out.
//
// Built with -ffp-contract=off: kept-box indices must be bit-exact.
#include "common.h"

namespace {

constexpr int NT = 256;          // threads per workgroup
constexpr int MAXN = 896;        // candidate capacity: sort+mask must fit 160 KiB LDS (decoder needs S*S*B <= 392)

struct Cand {
  float x1, y1, x2, y2, score;
  int cls;
  int src;  // index into the candidate list handed to NMS (decoder: compacted (i,j,b) order)
};

// shared layout (dynamic): cand[n] | sorted[n] | mask[n][W]  (W = ceil(n/64))
__device__ __forceinline__ bool suppressed_by(const Cand& a, float area_a, const Cand& b, float thr) {
  // reference utils.py:170-180 with a = the kept box i, b = a later box of `order[1:]`
  const float xx1 = fmaxf(b.x1, a.x1), yy1 = fmaxf(b.y1, a.y1);
  const float xx2 = fminf(b.x2, a.x2), yy2 = fminf(b.y2, a.y2);
  const float w = fmaxf(xx2 - xx1, 0.f), h = fmaxf(yy2 - yy1, 0.f);
  const float inter = w * h;
  const float area_b = (b.x2 - b.x1) * (b.y2 - b.y1);
  const float ovr = inter / ((area_a + area_b) - inter);
  return !(ovr <= thr);
}

// Sort + NMS over cand[0..n) held in LDS. On return keep_sorted[k] (k < *nkeep) holds sorted
// positions of kept boxes, in keep order.  All threads of the workgroup must call.
__device__ void sort_and_nms(const Cand* cand, Cand* sorted, unsigned long long* mask, int* keep_sorted,
                             int* nkeep, int n, float thr) {
  const int tid = threadIdx.x;
  const int W = (n + 63) >> 6;
  // rank sort: descending score, ties by ascending index (stable)
  for (int i = tid; i < n; i += NT) {
    const float s = cand[i].score;
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float sj = cand[j].score;
      rank += (sj > s) || (sj == s && j < i);
    }
    sorted[rank] = cand[i];
  }
  for (int i = tid; i < n * W; i += NT) mask[i] = 0ull;
  __syncthreads();
  // suppression matrix, upper triangle: one thread per (row r, word w)
  for (int idx = tid; idx < n * W; idx += NT) {
    const int r = idx / W, w = idx - r * W;
    const int c0 = w << 6;
    if (c0 + 63 <= r) continue;
    const Cand a = sorted[r];
    const float area_a = (a.x2 - a.x1) * (a.y2 - a.y1);
    unsigned long long bits = 0ull;
    const int cend = min(64, n - c0);
    for (int k = 0; k < cend; ++k) {
      const int c = c0 + k;
      if (c > r && suppressed_by(a, area_a, sorted[c], thr)) bits |= 1ull << k;
    }
    mask[idx] = bits;
  }
  __syncthreads();
  // sequential greedy walk by wavefront 0: lane w owns alive word w
  if (tid < 64) {
    const int lane = tid;
    unsigned long long alive = 0ull;
    if (lane < W) {
      const int rem = n - (lane << 6);
      alive = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
    }
    int kept = 0;
    for (int r = 0; r < n; ++r) {
      const unsigned long long word = __shfl(alive, r >> 6, 64);
      if ((word >> (r & 63)) & 1ull) {          // wave-uniform
        if (lane == 0) keep_sorted[kept] = r;
        ++kept;
        if (lane < W) alive &= ~mask[r * W + lane];
      }
    }
    if (lane == 0) *nkeep = kept;
  }
  __syncthreads();
}

struct DecodeArgs {
  const float* pred;   // [N,S,S,D] contiguous fp32
  int N, S, B, C;
  double thresh;       // compared in double, like the reference's Python float compare
  float nms_thr;
  int max_out;         // = S*S*B
  float* out_boxes;    // [N,max_out,4]
  long long* out_cls;  // [N,max_out]
  float* out_scores;   // [N,max_out]
  long long* out_keep; // [N,max_out] candidate indices, keep order
  int* out_counts;     // [N]
  int* out_ncand;      // [N] number of candidates before NMS (0 -> the zero box was substituted)
};

__global__ void __launch_bounds__(NT) k_decode_nms(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nslots = a.S * a.S * a.B;
  const int D = a.B * 5 + a.C;
  Cand* cand = (Cand*)smem_raw;
  Cand* sorted = cand + nslots;
  unsigned long long* mask = (unsigned long long*)(sorted + nslots);
  const int Wmax = (nslots + 63) >> 6;
  int* keep_sorted = (int*)(mask + (size_t)nslots * Wmax);
  int* misc = keep_sorted + nslots;       // [0]=nkeep [1]=ncand [2..2+4) wave counts, then float max
  float* fmisc = (float*)(misc + 8);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* p = a.pred + (size_t)blockIdx.x * a.S * a.S * D;

  // global max of the confidences (utils.py:113)
  float m = -INFINITY;
  for (int s = tid; s < nslots; s += NT) m = fmaxf(m, p[(s / a.B) * D + (s % a.B)]);
  m = wave_max(m);
  if (lane == 0) fmisc[wid] = m;
  __syncthreads();
  const float cmax = fmaxf(fmaxf(fmisc[0], fmisc[1]), fmaxf(fmisc[2], fmisc[3]));
  const float cell = (float)(1.0 / (double)a.S);      // Python 1./grid_num cast to fp32 by the tensor op
  __syncthreads();

  int base = 0;
  for (int s0 = 0; s0 < nslots; s0 += NT) {
    const int s = s0 + tid;
    bool ok = false;
    Cand c;
    if (s < nslots) {
      const int ci = s / a.B, b = s - ci * a.B;
      const int i = ci / a.S, j = ci - i * a.S;        // i = row (y), j = col (x)   utils.py:115-122
      const float* q = p + ci * D;
      const float conf = q[b];
      if (conf > 0.0001f || conf == cmax) {
        const float* bx = q + a.B + 4 * b;
        const float cx = bx[0] * cell + (float)j * cell;
        const float cy = bx[1] * cell + (float)i * cell;
        const float hw = 0.5f * bx[2], hh = 0.5f * bx[3];
        float best = q[5 * a.B];
        int bi = 0;
        for (int k = 1; k < a.C; ++k) {
          const float v = q[5 * a.B + k];
          if (v > best) { best = v; bi = k; }
        }
        const float score = conf * best;
        if ((double)score > a.thresh) {
          ok = true;
          c.x1 = cx - hw; c.y1 = cy - hh; c.x2 = cx + hw; c.y2 = cy + hh;
          c.score = score; c.cls = bi;
        }
      }
    }
    // order-preserving compaction across the 4 wavefronts
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) misc[2 + wid] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wid; ++w) off += misc[2 + w];
    if (ok) {
      const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
      c.src = pos;
      cand[pos] = c;
    }
    base += misc[2] + misc[3] + misc[4] + misc[5];
    __syncthreads();
  }
  int n = base;
  if (tid == 0) misc[1] = n;
  if (n == 0) {                       // reference substitutes one zero box, class 0, prob 0 (utils.py:134-137)
    if (tid == 0) { Cand z; z.x1 = z.y1 = z.x2 = z.y2 = z.score = 0.f; z.cls = 0; z.src = 0; cand[0] = z; }
    n = 1;
  }
  __syncthreads();
  sort_and_nms(cand, sorted, mask, keep_sorted, misc, n, a.nms_thr);
  const int nk = misc[0];
  const size_t ob = (size_t)blockIdx.x * a.max_out;
  for (int k = tid; k < nk; k += NT) {
    const Cand c = sorted[keep_sorted[k]];
    float* o = a.out_boxes + (ob + k) * 4;
    o[0] = c.x1; o[1] = c.y1; o[2] = c.x2; o[3] = c.y2;
    a.out_cls[ob + k] = c.cls;
    a.out_scores[ob + k] = c.score;
    a.out_keep[ob + k] = c.src;
  }
  if (tid == 0) { a.out_counts[blockIdx.x] = nk; a.out_ncand[blockIdx.x] = misc[1]; }
}

__global__ void __launch_bounds__(NT) k_nms(const float* __restrict__ boxes, const float* __restrict__ scores, int n,
                                            float thr, long long* out_keep, int* out_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Cand* cand = (Cand*)smem_raw;
  Cand* sorted = cand + n;
  unsigned long long* mask = (unsigned long long*)(sorted + n);
  const int W = (n + 63) >> 6;
  int* keep_sorted = (int*)(mask + (size_t)n * W);
  int* misc = keep_sorted + n;
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += NT) {
    Cand c;
    c.x1 = boxes[4 * i]; c.y1 = boxes[4 * i + 1]; c.x2 = boxes[4 * i + 2]; c.y2 = boxes[4 * i + 3];
    c.score = scores[i]; c.cls = 0; c.src = i;
    cand[i] = c;
  }
  __syncthreads();
  sort_and_nms(cand, sorted, mask, keep_sorted, misc, n, thr);
  const int nk = misc[0];
  for (int k = tid; k < nk; k += NT) out_keep[k] = sorted[keep_sorted[k]].src;
  if (tid == 0) *out_count = nk;
}

// pairwise IoU, utils/utils.py:10-57.  One thread per (n,m) pair.
__global__ void k_iou_matrix(const float* __restrict__ b1, const float* __restrict__ b2, int N, int M,
                             float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * M) return;
  const int n = idx / M, m = idx - n * M;
  const float* a = b1 + 4 * n;
  const float* b = b2 + 4 * m;
  float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
  float h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  w = w < 0.f ? 0.f : w;
  h = h < 0.f ? 0.f : h;
  const float I = w * h;
  const float a1 = (a[2] - a[0]) * (a[3] - a[1]);
  const float a2 = (b[2] - b[0]) * (b[3] - b[1]);
  out[idx] = I / ((a1 + a2) - I);
}

// utils/utils.py:59-75
__global__ void k_convert_cxcywh(const float* __restrict__ in, int n, float S, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float cx = in[4 * i] / S, cy = in[4 * i + 1] / S;
  const float hw = 0.5f * in[4 * i + 2], hh = 0.5f * in[4 * i + 3];
  out[4 * i] = cx - hw; out[4 * i + 1] = cy - hh; out[4 * i + 2] = cx + hw; out[4 * i + 3] = cy + hh;
}

size_t nms_lds_bytes(int n) {
  const size_t W = (n + 63) / 64;
  return (size_t)2 * n * sizeof(Cand) + (size_t)n * W * 8 + (size_t)n * 4 + 64;
}

}  // namespace

extern "C" int yv1_decode_nms_batched(const float* pred, int N, int S, int B, int C, double thresh, float nms_th,
                                      float* out_boxes, long long* out_cls, float* out_scores,
                                      long long* out_keep_idx, int* out_counts, int* out_ncand,
                                      hipStream_t stream) {
  if (!pred || !out_boxes || !out_cls || !out_scores || !out_keep_idx || !out_counts || !out_ncand)
    return YV1_ERR_BAD_ARG;
  if (N <= 0 || S <= 0 || B <= 0 || C <= 0) return YV1_ERR_BAD_ARG;
  const int nslots = S * S * B;
  if (nslots > MAXN) return YV1_ERR_UNSUPPORTED;
  DecodeArgs a;
  a.pred = pred; a.N = N; a.S = S; a.B = B; a.C = C; a.thresh = thresh; a.nms_thr = nms_th;
  a.max_out = nslots; a.out_boxes = out_boxes; a.out_cls = out_cls; a.out_scores = out_scores;
  a.out_keep = out_keep_idx; a.out_counts = out_counts; a.out_ncand = out_ncand;
  const size_t lds = nms_lds_bytes(nslots);
  static bool attr_set = false;
  if (!attr_set) {
    YV1_HIP(hipFuncSetAttribute((const void*)k_decode_nms, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_decode_nms, dim3(N), dim3(NT), lds, stream, a);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_nms(const float* boxes, const float* scores, int n, float threshold, long long* out_keep,
                       int* out_count, hipStream_t stream) {
  if (!out_keep || !out_count || n < 0) return YV1_ERR_BAD_ARG;
  if (n == 0) return (int)hipMemsetAsync(out_count, 0, sizeof(int), stream);
  if (!boxes || !scores) return YV1_ERR_BAD_ARG;
  if (n > MAXN) return YV1_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    YV1_HIP(hipFuncSetAttribute((const void*)k_nms, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_nms, dim3(1), dim3(NT), nms_lds_bytes(n), stream, boxes, scores, n, threshold, out_keep,
                     out_count);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_iou_matrix(const float* b1, int n, const float* b2, int m, float* out, hipStream_t stream) {
  if (n < 0 || m < 0) return YV1_ERR_BAD_ARG;
  if (n == 0 || m == 0) return YV1_OK;
  if (!b1 || !b2 || !out) return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_iou_matrix, dim3((n * m + 255) / 256), dim3(256), 0, stream, b1, b2, n, m, out);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}

extern "C" int yv1_convert_cxcywh_to_xyxy(const float* in, int n, int S, float* out, hipStream_t stream) {
  if (n < 0 || S <= 0) return YV1_ERR_BAD_ARG;
  if (n == 0) return YV1_OK;
  if (!in || !out) return YV1_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_convert_cxcywh, dim3((n + 255) / 256), dim3(256), 0, stream, in, n, (float)S, out);
  YV1_LAUNCH_CHECK();
  return YV1_OK;
}
