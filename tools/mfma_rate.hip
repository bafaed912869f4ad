// Development aid: issue rate of the bf16 MFMA shapes, in shader CYCLES (s_memtime stamps inside the kernel), with the
// MFMAs written in inline asm so that nothing but what is listed here sits between them, on random bf16 operands.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
// Round 2's version of this tool timed `__builtin_amdgcn_mfma_*` loops in ns at an assumed clock: hipcc rotated the eight
// accumulators through ~55 v_accvgpr_read/write per 8 MFMAs (see the listing), so its "13.2 ns per 16x16x32" measured the
// copies, not the matrix pipe.  This version prints cycles per MFMA and the clock the chip held (s_memrealtime = 100 MHz).
// FILL: single-issue instructions placed behind every MFMA (what a lone wave per SIMD can hide in an MFMA's shadow).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define M16(c) "v_mfma_f32_16x16x32_bf16 %" #c ", %8, %9, %" #c "\n\t"
#define M32(c) "v_mfma_f32_32x32x16_bf16 %" #c ", %8, %9, %" #c "\n\t"
#define F0
#define F1 "v_add_f32 %10, %10, %10\n\t"
#define F2 F1 "v_add_f32 %11, %11, %11\n\t"
#define F3 F2 "v_add_f32 %12, %12, %12\n\t"
#define F4 F3 "v_add_f32 %13, %13, %13\n\t"
#define F6 F4 "v_add_f32 %10, %10, %10\n\tv_add_f32 %11, %11, %11\n\t"
#define FE "v_exp_f32 %10, %10\n\t"                       // one transcendental
#define FEA "v_exp_f32 %10, %10\n\tv_add_f32 %11, %11, %11\n\t"
#define FS "s_nop 0\n\t"
#define BODY(M, F) M(0) F M(1) F M(2) F M(3) F M(4) F M(5) F M(6) F M(7) F

struct Out { unsigned long long cyc, rt; };

#define KERNEL(NAME, ACC, M, F)                                                                              \
  __global__ __launch_bounds__(256) void NAME(Out* out, float* sink, int iters) {                           \
    unsigned h = (threadIdx.x + blockIdx.x * 256u) * 2654435761u + 12345u;                                 \
    u32x4 a, b;                                                                                            \
    for (int i = 0; i < 4; ++i) {                                                                          \
      h = h * 1664525u + 1013904223u; a[i] = (h & 0x7fff7fffu) | 0x30003000u;                              \
      h = h * 1664525u + 1013904223u; b[i] = (h & 0x7fff7fffu) | 0x30003000u;                              \
    }                                                                                                      \
    ACC c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};                            \
    float v0 = threadIdx.x * 1e-3f, v1 = 0.5f, v2 = 0.25f, v3 = 0.125f;                                    \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");                                                         \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();     \
    asm volatile("s_waitcnt lgkmcnt(0)");                                                                  \
    for (int it = 0; it < iters; ++it) {                                                                   \
      asm volatile(BODY(M, F)                                                                              \
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)        \
                   : "v"(a), "v"(b), "v"(v0), "v"(v1), "v"(v2), "v"(v3));                                  \
      a[0] ^= 0x00550055u; b[1] ^= 0x002a002au;                                                                          \
    }                                                                                                      \
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");                                                         \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();     \
    asm volatile("s_waitcnt lgkmcnt(0)");                                                                  \
    if (threadIdx.x == 0) { out[blockIdx.x].cyc = t1 - t0; out[blockIdx.x].rt = r1 - r0; }                 \
    float s = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0] + v0 + v1 + v2 + v3;           \
    if (s == 12345.678f) sink[0] = s;                                                                      \
  }

KERNEL(k16_f0, f32x4, M16, F0)
KERNEL(k16_fs, f32x4, M16, FS)
KERNEL(k16_f1, f32x4, M16, F1)
KERNEL(k16_f2, f32x4, M16, F2)
KERNEL(k16_f3, f32x4, M16, F3)
KERNEL(k16_f4, f32x4, M16, F4)
KERNEL(k16_fe, f32x4, M16, FE)
KERNEL(k16_fea, f32x4, M16, FEA)
KERNEL(k32_f0, f32x16, M32, F0)
KERNEL(k32_f2, f32x16, M32, F2)
KERNEL(k32_f4, f32x16, M32, F4)
KERNEL(k32_f6, f32x16, M32, F6)

typedef void (*kern_t)(Out*, float*, int);
static void run(const char* name, kern_t k, int grid, int iters, double flop_per_mfma) {
  Out* d;
  float* sink;
  hipMalloc(&d, sizeof(Out) * grid);
  hipMalloc(&sink, 64);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, sink, 64);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, sink, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<Out> h(grid);
  hipMemcpy(h.data(), d, sizeof(Out) * grid, hipMemcpyDeviceToHost);
  std::vector<double> cyc(grid), clk(grid);
  for (int i = 0; i < grid; ++i) { cyc[i] = (double)h[i].cyc / (iters * 8.0); clk[i] = (double)h[i].cyc / (double)h[i].rt * 0.1; }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  const double tflops = flop_per_mfma * iters * 8.0 * 4 * grid / (ms * 1e-3) / 1e12;
  printf("%-28s grid %4d: %6.2f cycles per MFMA (median; min %.2f max %.2f), clock %.2f GHz, %.3f ms, %.0f TFLOP/s\n", name, grid,
         cyc[grid / 2], cyc[0], cyc[grid - 1], clk[grid / 2], ms, tflops);
  hipFree(d);
  hipFree(sink);
}

int main() {
  const int iters = 20000;
  const double f16 = 2.0 * 16 * 16 * 32, f32 = 2.0 * 32 * 32 * 16;
  for (int grid : {1, 256}) {
    run("16x16x32 bare", k16_f0, grid, iters, f16);
    run("16x16x32 + s_nop 0", k16_fs, grid, iters, f16);
    run("16x16x32 + 1 v_add", k16_f1, grid, iters, f16);
    run("16x16x32 + 2 v_add", k16_f2, grid, iters, f16);
    run("16x16x32 + 3 v_add", k16_f3, grid, iters, f16);
    run("16x16x32 + 4 v_add", k16_f4, grid, iters, f16);
    run("16x16x32 + 1 v_exp", k16_fe, grid, iters, f16);
    run("16x16x32 + v_exp + v_add", k16_fea, grid, iters, f16);
    run("32x32x16 bare", k32_f0, grid, iters, f32);
    run("32x32x16 + 2 v_add", k32_f2, grid, iters, f32);
    run("32x32x16 + 4 v_add", k32_f4, grid, iters, f32);
    run("32x32x16 + 6 v_add", k32_f6, grid, iters, f32);
  }
  return 0;
}
