// Feasibility probe for the 32 x 32 x 16 redesign of the decoder recurrences: can one wave hold the weights of
// 32 units x 3 gates x K = 512 (96 fragments = 384 registers) beside three 32 x 32 accumulators, and how long does the MFMA
// phase of a 32-row tile take with the activation fragments read from LDS?
//   hipcc -O3 --offload-arch=gfx950 tools/proto32.hip -o tools/proto32.bin && tools/proto32.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int KS = 32;   // k-steps of 16 over K = 512

__device__ __forceinline__ f32x16 mma32a(const bf16x8& w, const bf16x8& x, f32x16 c) {
  asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(w), "v"(x));
  return c;
}
__device__ __forceinline__ f32x16 mma32(const bf16x8& w, const bf16x8& x, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, c, 0, 0, 0);
}

// LDS tile: 32 rows x 512 bf16 (1 KB per row), 16-byte chunk q of row r at position q ^ (r & 31)
__global__ __launch_bounds__(256) void proto_kernel(const __bf16* W, float* out, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, kq = lane >> 5;            // sequence column, k half of the 16-wide step
  bf16x8 w0[KS], w1[KS], w2[KS];                      // gate r, z: AGPR candidates; gate n: VGPRs
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const __bf16* p = W + ((long)(wave * 3) * 32 + n) * 512 + ks * 16 + kq * 8;
    w0[ks] = *reinterpret_cast<const bf16x8*>(p);
    w1[ks] = *reinterpret_cast<const bf16x8*>(p + 32 * 512);
    w2[ks] = *reinterpret_cast<const bf16x8*>(p + 64 * 512);
  }
  for (int i = tid; i < 32 * 1024 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c013c00u, 0x3c003c02u);
  __syncthreads();
  float tot = 0.f;
  const char* A = smem + n * 1024;
  for (int t = 0; t < tiles; ++t) {
    f32x16 c0, c1, c2;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; }
#pragma unroll
    for (int g4 = 0; g4 < KS / 4; ++g4) {
      bf16x8 x[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ks = g4 * 4 + i;
        x[i] = *reinterpret_cast<const bf16x8*>(A + ((((ks * 2 + kq) ^ n) & 63) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ks = g4 * 4 + i;
        c0 = mma32a(w0[ks], x[i], c0);
        c1 = mma32a(w1[ks], x[i], c1);
        c2 = mma32(w2[ks], x[i], c2);
      }
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2));
    for (int e = 0; e < 16; ++e) tot += c0[e] * 0.5f + c1[e] + c2[e] * 0.25f;
    __builtin_amdgcn_s_barrier();
  }
  out[blockIdx.x * 256 + tid] = tot;
}

int main() {
  __bf16* W;
  float* out;
  hipMalloc(&W, 4 * 96 * 512 * 2);
  hipMemset(W, 0x3c, 4 * 96 * 512 * 2);
  hipMalloc(&out, 256 * 256 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(proto_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024);
  for (int grid : {1, 240}) {
    hipLaunchKernelGGL(proto_kernel, dim3(grid), dim3(256), 32 * 1024, 0, W, out, 16);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int tiles = 4000;
    hipEventRecord(e0);
    hipLaunchKernelGGL(proto_kernel, dim3(grid), dim3(256), 32 * 1024, 0, W, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("grid %3d: %.3f us per 32-row tile (96 MFMAs of 32x32x16 per wave, fragments from LDS), %.1f TFLOP/s\n", grid, ms * 1e3 / tiles,
           2.0 * 32 * 32 * 16 * 96 * 4 * grid * tiles / (ms * 1e-3) / 1e12);
  }
  return 0;
}
