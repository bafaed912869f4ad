// Development aid: sustained issue rate of the bf16 MFMA instructions on one CU and on the whole chip.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
// Every wave runs ITER x 8 independent MFMAs (no memory traffic); time / (ITER * 8 * waves per SIMD) = ns per MFMA and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float s = 0.f;
  if constexpr (KIND == 0) {
    f32x4 c[8];
    for (int k = 0; k < 8; ++k) c[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k], 0, 0, 0);
    }
    for (int k = 0; k < 8; ++k) s += c[k][0];
  } else if constexpr (KIND == 2) {   // the gfx940-era half-K shape
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    s16x4 a4, b4;
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(i * 3 + 1); }
    f32x4 c[8];
    for (int k = 0; k < 8; ++k) c[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c[k], 0, 0, 0);
    }
    for (int k = 0; k < 8; ++k) s += c[k][0];
  } else if constexpr (KIND == 20) {
    f32x16 c[8];
    for (int k = 0; k < 8; ++k)
      for (int e = 0; e < 16; ++e) c[k][e] = 0.f;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 ra, rb;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 4; ++i) { h = h * 1664525u + 1013904223u; ra[i] = (h & 0x7fff7fffu) | 0x30003000u; h = h * 1664525u + 1013904223u; rb[i] = (h & 0x7fff7fffu) | 0x30003000u; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, rb), c[k], 0, 0, 0);
        ra[k & 3] ^= 0x00550055u << (k & 7);      // one VALU per MFMA: operands keep changing
      }
    }
    for (int k = 0; k < 8; ++k) s += c[k][0];
  } else if constexpr (KIND >= 10) {   // 16x16x32 with a filler between MFMAs
    f32x4 c[8];
    for (int k = 0; k < 8; ++k) c[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k], 0, 0, 0);
        if constexpr (KIND == 10) asm volatile("s_nop 0");
        if constexpr (KIND == 11) asm volatile("s_nop 1");
        if constexpr (KIND == 12) asm volatile("s_nop 3");
        if constexpr (KIND == 13) asm volatile("s_nop 7");
        if constexpr (KIND == 14) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
        if constexpr (KIND == 15) asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0" : "+v"(v));
        if constexpr (KIND == 16) asm volatile("s_nop 0\n\ts_nop 0");
        if constexpr (KIND == 17) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
      }
    }
    for (int k = 0; k < 8; ++k) s += c[k][0];
    s += v;
  } else if constexpr (KIND == 3) {   // 16x16x32 with a VALU instruction between MFMAs (is the gap an issue bubble?)
    f32x4 c[8];
    for (int k = 0; k < 8; ++k) c[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k], 0, 0, 0);
        asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0" : "+v"(v));
      }
    }
    for (int k = 0; k < 8; ++k) s += c[k][0];
    s += v;
  } else {
    f32x16 c[8];
    for (int k = 0; k < 8; ++k)
      for (int e = 0; e < 16; ++e) c[k][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[k], 0, 0, 0);
    }
    for (int k = 0; k < 8; ++k) s += c[k][0];
  }
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
static void run(const char* name, int grid, int block, int iters) {
  float* d;
  hipMalloc(&d, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(block), 0, 0, d, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(block), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int waves_per_simd = block / 64 / 4 > 0 ? block / 64 / 4 : 1;
  const double n_per_simd = (double)iters * 8 * waves_per_simd;
  const double flop = (KIND == 2 ? 2.0 * 16 * 16 * 16 : (KIND == 1 || KIND == 20) ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32) * iters * 8.0 * (block / 64) * grid;
  printf("%-22s grid %4d block %4d: %.3f ms, %.2f ns per MFMA and SIMD, %.1f TFLOP/s\n", name, grid, block, ms,
         ms * 1e6 / n_per_simd, flop / (ms * 1e-3) / 1e12);
  hipFree(d);
}

int main() {
  const int iters = 20000;
  run<10>("16x16x32 + s_nop 0", 1, 256, iters);
  run<11>("16x16x32 + s_nop 1", 1, 256, iters);
  run<12>("16x16x32 + s_nop 3", 1, 256, iters);
  run<13>("16x16x32 + s_nop 7", 1, 256, iters);
  run<14>("16x16x32 + 1 VALU", 1, 256, iters);
  run<15>("16x16x32 + 4 VALU", 1, 256, iters);
  run<16>("16x16x32 + 2x s_nop 0", 1, 256, iters);
  run<17>("16x16x32 + 1 SALU", 1, 256, iters);
  run<2>("16x16x16 bf16 (1k)", 1, 256, iters);
  run<3>("16x16x32 bf16 + 2 VALU", 1, 256, iters);
  run<0>("16x16x32 bf16", 1, 256, iters);      // one workgroup: one wave per SIMD of one CU
  run<1>("32x32x16 bf16", 1, 256, iters);
  run<0>("16x16x32 bf16", 256, 256, iters);    // one wave per SIMD on every CU
  run<1>("32x32x16 bf16", 256, 256, iters);
  run<20>("32x32x16 random data", 1, 256, iters);
  run<20>("32x32x16 random data", 256, 256, iters);
  run<20>("32x32x16 random data", 192, 256, iters);
  run<0>("16x16x32 bf16", 1024, 256, iters);   // four workgroups per CU = four waves per SIMD: divide the ns by 4
  run<1>("32x32x16 bf16", 1024, 256, iters);
  return 0;
}
