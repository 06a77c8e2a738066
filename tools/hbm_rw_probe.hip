// Pure write / pure read / copy stream rates of one MI355X with 16-byte-per-lane accesses (what the elementwise kernels
// and the conv epilogue issue).  hipcc --offload-arch=gfx950 tools/hbm_rw_probe.hip -o /tmp/hbm_rw_probe && /tmp/hbm_rw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_write(uint4* p, size_t n) {
  const uint4 v = make_uint4(1, 2, 3, 4);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_read(const uint4* p, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_copy(const uint4* a, uint4* b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
  const size_t bytes = 411041792, n = bytes / 16;   // 64 x 112 x 112 x 256 bf16
  uint4 *a, *b; unsigned* o;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
  (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int blocks : {2048, 8192, 16384}) {
    float ms;
    for (int mode = 0; mode < 3; ++mode) {
      for (int it = 0; it < 3; ++it) {
        if (mode == 0) k_write<<<blocks, 256>>>(a, n); else if (mode == 1) k_read<<<blocks, 256>>>(a, n, o); else k_copy<<<blocks, 256>>>(a, b, n);
      }
      (void)hipEventRecord(e0);
      for (int it = 0; it < 10; ++it) {
        if (mode == 0) k_write<<<blocks, 256>>>(a, n); else if (mode == 1) k_read<<<blocks, 256>>>(a, n, o); else k_copy<<<blocks, 256>>>(a, b, n);
      }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 100.0, gb = (mode == 2 ? 2.0 : 1.0) * bytes / 1e9;
      printf("blocks %5d  %-5s %7.1f us  %5.2f TB/s\n", blocks, mode == 0 ? "write" : mode == 1 ? "read" : "copy", us, gb / us * 1e3);
    }
  }
  return 0;
}
