// Feasibility probe for the 32 x 32 x 16 redesign of the decoder recurrences.  One workgroup = 64 units of a
// GRU layer on a 32-row tile: waves (ub, kh) = unit block of 32 x {h half, x half}.  Every wave keeps the weights of its three
// products in registers (3 gates x 32 units x K = 512: 96 fragments = 384 registers), runs 96 MFMAs per tile with the
// activation fragments read from a swizzled LDS tile, hands half of its accumulators to its partner through LDS and runs the
// gate arithmetic on the other half.  Measures registers (listing) and time per tile; no global traffic in the loop.
// PROBE_MFMA_ONLY=1 at compile time: the MFMA phase alone.
//   hipcc -O3 --offload-arch=gfx950 tools/proto32.hip -o tools/proto32.bin && tools/proto32.bin
// Measured (one CU / 240 CUs): MFMA phase alone 1.64 / 2.06 us per 32-row tile, 504 registers; with the exchange and the
// gate arithmetic 3.18 / 3.39 us, 512 registers, no spills (16-byte exchange accesses: 8 spills, 3.7 us).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef PROBE_MFMA_ONLY
#define PROBE_MFMA_ONLY 0
#endif
constexpr int KS = 32;                       // k-steps of 16 over K = 512
constexpr int TILE_BYTES = 32 * 1024;        // 32 rows x 512 bf16 per operand
constexpr int XCH_BYTES = 4 * 3 * 8 * 256;   // per wave 3 accumulators x 8 registers x 64 lanes x 4 B
constexpr int SMEM = 2 * TILE_BYTES + XCH_BYTES + 32 * 64 * 2;

__device__ __forceinline__ void mma4(f32x16& c, const bf16x8& w0, const bf16x8& w1, const bf16x8& w2, const bf16x8& w3,
                                     const bf16x8& x0, const bf16x8& x1, const bf16x8& x2, const bf16x8& x3) {
  asm("v_mfma_f32_32x32x16_bf16 %0, %1, %5, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %2, %6, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %3, %7, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %4, %8, %0"
      : "+v"(c) : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
}
__device__ __forceinline__ void mma4v(f32x16& c, const bf16x8& w0, const bf16x8& w1, const bf16x8& w2, const bf16x8& w3,
                                      const bf16x8& x0, const bf16x8& x1, const bf16x8& x2, const bf16x8& x3) {
  asm("v_mfma_f32_32x32x16_bf16 %0, %1, %5, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %2, %6, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %3, %7, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %4, %8, %0"
      : "+v"(c) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
}
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 2.0f * sigm(2.0f * x) - 1.0f; }

__global__ __launch_bounds__(256) void proto_kernel(const __bf16* W, float* out, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ub = wave & 1, kh = wave >> 1;           // unit block, K half (0: h operand, 1: x operand)
  const int n = lane & 31, kq = lane >> 5;           // sequence column, k half of the 16-wide step
  bf16x8 w0[KS], w1[KS], w2[KS];                     // gates r, z (AGPR candidates) and hn / in
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const __bf16* p = W + ((long)(wave * 3) * 32 + n) * 512 + ks * 16 + kq * 8;
    w0[ks] = *reinterpret_cast<const bf16x8*>(p);
    w1[ks] = *reinterpret_cast<const bf16x8*>(p + 32 * 512);
    w2[ks] = *reinterpret_cast<const bf16x8*>(p + 64 * 512);
  }
  for (int i = tid; i < 2 * TILE_BYTES / 16; i += 256)
    reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + (i & 255), 0x3c003c00u, 0x3c013c00u, 0x3c003c02u);
  __syncthreads();
  char* sX = smem + 2 * TILE_BYTES;                  // exchange: [wave][acc 3][reg 8][lane 64] fp32
  char* sH = sX + XCH_BYTES;                         // output staging [32 rows][64 units] bf16
  float hprev[8];
  for (int e = 0; e < 8; ++e) hprev[e] = 0.01f * e;
  float tot = 0.f;
  const char* A = smem + kh * TILE_BYTES + n * 1024;
  for (int t = 0; t < tiles; ++t) {
    f32x16 c0, c1, c2;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; }
#pragma unroll
    for (int g4 = 0; g4 < KS / 4; ++g4) {
      bf16x8 x[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const bf16x8*>(A + (((((g4 * 4 + i) * 2 + kq) ^ n) & 63) << 4));
      mma4(c0, w0[g4 * 4], w0[g4 * 4 + 1], w0[g4 * 4 + 2], w0[g4 * 4 + 3], x[0], x[1], x[2], x[3]);
      mma4(c1, w1[g4 * 4], w1[g4 * 4 + 1], w1[g4 * 4 + 2], w1[g4 * 4 + 3], x[0], x[1], x[2], x[3]);
      mma4v(c2, w2[g4 * 4], w2[g4 * 4 + 1], w2[g4 * 4 + 2], w2[g4 * 4 + 3], x[0], x[1], x[2], x[3]);
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2));
    if (PROBE_MFMA_ONLY) {
      for (int e = 0; e < 16; ++e) tot += c0[e] * 0.5f + c1[e] + c2[e] * 0.25f;
      __builtin_amdgcn_s_barrier();
      continue;
    }
    // hand the half of the accumulators that the partner finalises to it: kh 0 keeps registers 0..7, kh 1 keeps 8..15
    float* mine = reinterpret_cast<float*>(sX) + (wave * 3 * 8) * 64 + lane;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int src = kh ? e : 8 + e;
      mine[(0 * 8 + e) * 64] = c0[src];
      mine[(1 * 8 + e) * 64] = c1[src];
      mine[(2 * 8 + e) * 64] = c2[src];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float* theirs = reinterpret_cast<const float*>(sX) + ((wave ^ 2) * 3 * 8) * 64 + lane;
    float hn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int own = kh ? 8 + e : e;
      const float pr = c0[own] + theirs[(0 * 8 + e) * 64];
      const float pz = c1[own] + theirs[(1 * 8 + e) * 64];
      const float a = c2[own], b = theirs[(2 * 8 + e) * 64];     // hn and in: one from each half
      const float r = sigm(pr), z = sigm(pz);
      const float nn = tanh_((kh ? a : b) + r * (kh ? b : a));
      hn[e] = (1.0f - z) * nn + z * hprev[e];
      hprev[e] = hn[e];
    }
    // staging: lane = sequence column n; the wave's 8 units of the block: rows of 4 consecutive units
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      bf16x4 o;
      for (int i = 0; i < 4; ++i) o[i] = (__bf16)hn[q * 4 + i];
      const int unit = ub * 32 + kh * 16 + q * 8 + kq * 4;
      *reinterpret_cast<bf16x4*>(sH + n * 128 + unit * 2) = o;
    }
    for (int e = 0; e < 8; ++e) tot += hn[e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
  hipFuncSetAttribute(reinterpret_cast<const void*>(proto_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  for (int grid : {1, 240}) {
    hipLaunchKernelGGL(proto_kernel, dim3(grid), dim3(256), SMEM, 0, W, out, 16);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int tiles = 4000;
    hipEventRecord(e0);
    hipLaunchKernelGGL(proto_kernel, dim3(grid), dim3(256), SMEM, 0, W, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("grid %3d: %.3f us per 32-row tile (96 MFMAs of 32x32x16 per wave%s), %.1f TFLOP/s\n", grid, ms * 1e3 / tiles,
           PROBE_MFMA_ONLY ? "" : " + exchange + gate arithmetic", 2.0 * 32 * 32 * 16 * 96 * 4 * grid * tiles / (ms * 1e-3) / 1e12);
  }
  return 0;
}
