// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of the YOLO-v1 hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define YV1_OK 0
#define YV1_ERR_BAD_ARG 1001
#define YV1_ERR_UNSUPPORTED 1002
#define YV1_ERR_WORKSPACE 1003

#define YV1_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

#define YV1_HIP(call)                              \
  do {                                             \
    hipError_t e__ = (call);                       \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

// Raises a kernel's dynamic-LDS limit once PER DEVICE (the attribute is per device, a process may drive several) and
// thread-safely: one bit per device ordinal in an atomic latch that belongs to the expansion site (= one kernel
// instantiation).  Two threads racing here both set the same value, which is harmless.
#define YV1_SET_MAX_LDS(kern, bytes)                                                                              \
  do {                                                                                                            \
    static std::atomic<unsigned long long> done__{0ull};                                                          \
    int dev__ = 0;                                                                                                \
    YV1_HIP(hipGetDevice(&dev__));                                                                                \
    const unsigned long long bit__ = 1ull << (dev__ & 63);                                                        \
    if (!(done__.load(std::memory_order_acquire) & bit__)) {                                                      \
      YV1_HIP(hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      done__.fetch_or(bit__, std::memory_order_release);                                                          \
    }                                                                                                             \
  } while (0)

// dispatch record (cfglog.hip): every convolution entry point resets it, every kernel launch appends its template name
void yv1_cfg_reset();
void yv1_cfg_note(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

typedef unsigned short bf16_t;  // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even; NaN stays NaN (plain cast path, see MI355X_MICROARCH "Correctness boundaries")
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two fp32 -> packed bf16 pair in one v_cvt_pk_bf16_f32 (gfx950): round to nearest even, NaN stays NaN -- the same
// values as f32_to_bf16 above at a sixth of the VALU work (the elementwise kernels convert 8 values per 16-byte store)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float yv1_f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 yv1_bf16x2;
  const yv1_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, yv1_bf16x2));
}

// four fp32 -> four OCP e4m3 bytes (round to nearest even; inputs clamped to the finite range +-448 first)
__device__ __forceinline__ unsigned pack_e4m3x4(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

// 64-lane wavefront reductions (CDNA wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
