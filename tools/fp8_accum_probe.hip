// How exactly does v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3, unit block scales) add its 64 products?
// The fp8 parity tests compare the kernels with fp32 F.conv2d on the same e4m3 operands; every product of two e4m3 values
// is exact in fp32, so the only difference is how the sum is formed.  At batch 64 (2e8 outputs per layer) a few hundred
// outputs sat further from the fp32 reference than fp32 re-association explains (tests/test_gpu_bench_configs_fp8.py).
// This probe feeds one big product and 63 equal small ones (operand bytes paired per lane, so the k <-> byte map does not
// matter) and prints, per exponent gap, the MFMA result against the exactly rounded sum:
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_accum_probe.hip -o tools/bin/fp8_accum_probe && tools/bin/fp8_accum_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

__global__ void k_scaled(const unsigned char* a, const unsigned char* b, const float* cin, float* out) {
  const int lane = threadIdx.x;
  i32x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = ((const int*)(a + lane * 32))[i]; vb[i] = ((const int*)(b + lane * 32))[i]; }
  f32x16 c;
  for (int e = 0; e < 16; ++e) c[e] = cin[0];
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int e = 0; e < 16; ++e) out[lane * 16 + e] = c[e];
}

static unsigned char e4m3_pow2(int e) {          // 2^e, e in [-9, 8]
  if (e >= -6) return (unsigned char)((e + 7) << 3);
  return (unsigned char)(1 << (e + 9));          // denormals 2^-9, 2^-8, 2^-7
}

int main() {
  unsigned char ha[64 * 32], hb[64 * 32], *da, *db;
  float hout[1024], *dout, *dc, hc;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, sizeof hout); hipMalloc(&dc, 4);
  printf("one product 2^8 (16x16) + 63 products 2^-s each, C = 0:   exact = 256 + 63*2^-s\n");
  printf("%4s %22s %22s %12s\n", "s", "mfma", "exact (fp32 RNE)", "diff/2^8");
  for (int s = 0; s <= 18; ++s) {
    // small product 2^-s = 2^ea * 2^eb with ea, eb in [-9, 0]
    int ea = -(s / 2), eb = -(s - s / 2);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
      const bool big = (l >> 5) == 0 && j == 0;            // one byte position of the first half: one k per row / column
      ha[l * 32 + j] = big ? e4m3_pow2(4) : e4m3_pow2(ea);
      hb[l * 32 + j] = big ? e4m3_pow2(4) : e4m3_pow2(eb);
    }
    hc = 0.f;
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dc, &hc, 4, hipMemcpyHostToDevice);
    k_scaled<<<1, 64>>>(da, db, dc, dout);
    hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
    const double exact = 256.0 + 63.0 * ldexp(1.0, -s);
    const float want = (float)exact;
    printf("%4d %22.12f %22.12f %12.3e\n", s, hout[0], want, (hout[0] - exact) / 256.0);
  }
  printf("\nC = 2^8 (accumulator input), 64 products 2^-s each:   exact = 256 + 64*2^-s\n");
  for (int s = 4; s <= 18; s += 2) {
    int ea = -(s / 2), eb = -(s - s / 2);
    for (int i = 0; i < 64 * 32; ++i) { ha[i] = e4m3_pow2(ea); hb[i] = e4m3_pow2(eb); }
    hc = 256.f;
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dc, &hc, 4, hipMemcpyHostToDevice);
    k_scaled<<<1, 64>>>(da, db, dc, dout);
    hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
    const double exact = 256.0 + 64.0 * ldexp(1.0, -s);
    printf("%4d %22.12f %22.12f %12.3e\n", s, hout[0], (float)exact, (hout[0] - exact) / 256.0);
  }
  printf("\nmixed signs: +2^8, -2^8, then 62 products 2^-s:   exact = 62*2^-s\n");
  for (int s = 4; s <= 18; s += 2) {
    int ea = -(s / 2), eb = -(s - s / 2);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
      const bool h0 = (l >> 5) == 0;
      unsigned char av = e4m3_pow2(ea), bv = e4m3_pow2(eb);
      if (h0 && j == 0) { av = e4m3_pow2(4); bv = e4m3_pow2(4); }
      if (h0 && j == 1) { av = e4m3_pow2(4) | 0x80; bv = e4m3_pow2(4); }
      ha[l * 32 + j] = av; hb[l * 32 + j] = bv;
    }
    hc = 0.f;
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dc, &hc, 4, hipMemcpyHostToDevice);
    k_scaled<<<1, 64>>>(da, db, dc, dout);
    hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
    const double exact = 62.0 * ldexp(1.0, -s);
    printf("%4d %22.12f %22.12f %12.3e (of 2^8)\n", s, hout[0], (float)exact, (hout[0] - exact) / 256.0);
  }
  return 0;
}
