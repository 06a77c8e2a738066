// Probe: does gfx950 execute scalar-cache atomics (s_atomic_add, returned value in an SGPR, counted by lgkmcnt)?
// 2048 workgroups x 4 waves; wave 0 of each takes 3 tickets; all tickets must be distinct and cover [0, 3*2048).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
__global__ void k(unsigned* ctr, unsigned* out) {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid == 0) {
    for (int i = 0; i < 3; ++i) {
      unsigned v = 1;
      asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
      if (threadIdx.x == 0) out[blockIdx.x * 3 + i] = v;
    }
  }
}
int main() {
  unsigned *ctr, *out;
  const int G = 2048;
  hipMalloc(&ctr, 4); hipMalloc(&out, G * 3 * 4);
  hipMemset(ctr, 0, 4);
  hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, ctr, out);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 1;
  std::vector<unsigned> h(G * 3);
  unsigned c = 0;
  hipMemcpy(h.data(), out, G * 3 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool ok = c == (unsigned)G * 3;
  for (int i = 0; i < G * 3; ++i) ok = ok && h[i] == (unsigned)i;
  printf("counter %u, tickets %s\n", c, ok ? "distinct and complete" : "WRONG");
  return ok ? 0 : 2;
}
