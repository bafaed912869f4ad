// Development aid: which SIMD does each wave of a 512-thread (and 256-thread) workgroup land on?  (HW_REG_HW_ID bits 5:4)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 4096 * 4);
  for (int bs : {512, 256}) {
    const int nb = 16, nw = bs / 64;
    hipLaunchKernelGGL(probe, dim3(nb), dim3(bs), 100 * 1024, 0, d);   // big LDS request: one workgroup per CU
    unsigned h[4096];
    hipMemcpy(h, d, nb * nw * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < nb; ++b) {
      printf("block %2d (%d threads): simd of waves:", b, bs);
      for (int w = 0; w < nw; ++w) printf(" %u", (h[b * nw + w] >> 4) & 3);
      printf("   cu %u wave slots:", (h[b * nw] >> 8) & 15);
      for (int w = 0; w < nw; ++w) printf(" %u", h[b * nw + w] & 15);
      printf("\n");
    }
  }
  return 0;
}
