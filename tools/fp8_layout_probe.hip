#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// out[row][col] = sum_k A[row][k]*B[k][col]; we feed per-lane operand bytes from global arrays indexed [lane][byte]
__global__ void k_scaled(const unsigned char* a, const unsigned char* b, float* out) {
  const int lane = threadIdx.x;
  i32x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = ((const int*)(a + lane * 32))[i]; vb[i] = ((const int*)(b + lane * 32))[i]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int e = 0; e < 16; ++e) {
    const int col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    out[row * 32 + col] = c[e];
  }
}
__global__ void k_plain(const unsigned char* a, const unsigned char* b, float* out) {
  const int lane = threadIdx.x;
  long va = *(const long*)(a + lane * 8), vb = *(const long*)(b + lane * 8);
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(va, vb, c, 0, 0, 0);
  for (int e = 0; e < 16; ++e) {
    const int col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    out[row * 32 + col] = c[e];
  }
}
// e4m3 encode of small ints: 0,1,2,..: use table from host
static unsigned char e4m3(float v) {  // exact for small integers / powers of two
  if (v == 0) return 0;
  int s = v < 0; v = fabsf(v);
  int e = (int)floorf(log2f(v)); float m = v / exp2f(e) - 1.f;
  int mi = (int)roundf(m * 8); if (mi == 8) { mi = 0; e++; }
  return (unsigned char)((s << 7) | ((e + 7) << 3) | mi);
}
int main() {
  // hypothesis H(scaled): lane l (r=l&31,h=l>>5) byte j holds A[r][k=32h+j]; B[k=32h+j][col r]
  // test: A[r][k] = (r%7)+1 if ... use random small ints in {-3..3}; B likewise; compare with CPU for the hypothesis.
  const int K = 64;
  float A[32][64], B[64][32];
  srand(1);
  for (int r = 0; r < 32; ++r) for (int k = 0; k < K; ++k) { A[r][k] = (float)(rand() % 7 - 3); B[k][r] = (float)(rand() % 5 - 2); }
  unsigned char ha[64 * 32], hb[64 * 32];
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
    int r = l & 31, h = l >> 5, k = 32 * h + j;
    ha[l * 32 + j] = e4m3(A[r][k]); hb[l * 32 + j] = e4m3(B[k][r]);
  }
  unsigned char *da, *db; float* dout; float hout[1024];
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, sizeof hout);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  k_scaled<<<1, 64>>>(da, db, dout);
  hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { float s = 0; for (int k = 0; k < K; ++k) s += A[r][k] * B[k][c]; if (s != hout[r * 32 + c]) ++bad; }
  printf("scaled 32x32x64 hypothesis k=32h+j: %d mismatches (sample %f)\n", bad, hout[5]);
  // alternative hypothesis: k = 16*(j/16)*2 ... try k = 16h + (j%16) + 32*(j/16)
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
    int r = l & 31, h = l >> 5, k = 16 * h + (j % 16) + 32 * (j / 16);
    ha[l * 32 + j] = e4m3(A[r][k]); hb[l * 32 + j] = e4m3(B[k][r]);
  }
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  k_scaled<<<1, 64>>>(da, db, dout);
  hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
  bad = 0;
  for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { float s = 0; for (int k = 0; k < K; ++k) s += A[r][k] * B[k][c]; if (s != hout[r * 32 + c]) ++bad; }
  printf("scaled 32x32x64 hypothesis k=16h+(j%%16)+32(j/16): %d mismatches\n", bad);
  // plain 32x32x16 fp8: k = 8h + j
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
    int r = l & 31, h = l >> 5, k = 8 * h + j;
    ha[l * 8 + j] = e4m3(A[r][k]); hb[l * 8 + j] = e4m3(B[k][r]);
  }
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  k_plain<<<1, 64>>>(da, db, dout);
  hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
  bad = 0;
  for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { float s = 0; for (int k = 0; k < 16; ++k) s += A[r][k] * B[k][c]; if (s != hout[r * 32 + c]) ++bad; }
  printf("plain 32x32x16 fp8 hypothesis k=8h+j: %d mismatches\n", bad);
  // value checks: 448 and a denormal
  return 0;
}
