// Common device/host helpers for the POLYGON SMILES-VAE HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// ---- error plumbing (no C++ exceptions cross the C ABI) -----------------------------------
extern thread_local char g_pg_err[512];
#define PG_OK 0
#define PG_ERR_ARG (-1)
#define PG_ERR_HIP (-2)
#define PG_ERR_UNSUPPORTED (-3)
#define PG_ERR_WORKSPACE (-4)
#define PG_ERR_PERSIST (-5)

#define PG_FAIL(code, ...)                          \
  do {                                              \
    snprintf(g_pg_err, sizeof(g_pg_err), __VA_ARGS__); \
    return (code);                                  \
  } while (0)

#define PG_CHECK_HIP(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) PG_FAIL(PG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define PG_CHECK_LAUNCH() PG_CHECK_HIP(hipGetLastError())

#define PG_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) PG_FAIL(PG_ERR_ARG, __VA_ARGS__); \
  } while (0)

// ---- scalar conversions -------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// v_rcp_f32 (1 ulp) instead of an IEEE division: the gate epilogues run on ONE wave per SIMD in the persistent kernels,
// where every VALU instruction is exposed
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); exact limits at +-inf, abs error ~2e-7
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ---- Philox4x32-10 (counter-based RNG; same stream whatever the launch geometry) ------------
struct Philox4 {
  uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ uint32_t pg_mulhi32(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint32_t hi0 = pg_mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = pg_mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox4 r = {c0, c1, c2, c3};
  return r;
}
// uniform in [0,1) with 24 bits
__host__ __device__ __forceinline__ float pg_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// dropout keep/scale factors of the 4 consecutive columns 4*col4 .. 4*col4+3 of (layer, t, global batch row); p = drop
// probability.  Every kernel (forward mask, BPTT mask) goes through these two, so the mask is a pure function of its
// coordinates whatever the launch geometry.  A counter hash (three rounds of the lowbias32 integer mixer: 6 integer
// multiplies for 4 masks) rather than Philox-10 (40 quarter-rate multiplies): the mask is recomputed in the gate epilogue
// of every cell, forward and backward.  Each mask compares a 16-bit uniform with round(p * 65536).
__host__ __device__ __forceinline__ uint32_t pg_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ void dropout_scale4(uint64_t seed, int layer, int t, int bglob, int col4, float p, float inv_keep,
                                               float (&m)[4]) {
  const uint32_t rowkey = pg_mix32((uint32_t)seed ^ ((uint32_t)bglob * 0x9E3779B1u) ^ ((uint32_t)t * 0x85EBCA77u) ^
                                   ((uint32_t)layer * 0xC2B2AE3Du) ^ (uint32_t)(seed >> 32) * 0x27D4EB2Fu);
  const uint32_t w0 = pg_mix32(rowkey + (uint32_t)col4 * 0x9E3779B9u);
  const uint32_t w1 = pg_mix32(w0 ^ 0x68E31DA4u);
  const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
  m[0] = (w0 & 0xFFFFu) >= thr ? inv_keep : 0.0f;
  m[1] = (w0 >> 16) >= thr ? inv_keep : 0.0f;
  m[2] = (w1 & 0xFFFFu) >= thr ? inv_keep : 0.0f;
  m[3] = (w1 >> 16) >= thr ? inv_keep : 0.0f;
}
__device__ __forceinline__ float dropout_scale(uint64_t seed, int layer, int t, int bglob, int col, float p, float inv_keep) {
  float m[4];
  dropout_scale4(seed, layer, t, bglob, col >> 2, p, inv_keep, m);
  const int e = col & 3;
  return e == 0 ? m[0] : e == 1 ? m[1] : e == 2 ? m[2] : m[3];
}

// LDS requests above 64 KiB need the per-function opt-in (once per kernel per process).
#define PG_LAUNCH_BIG_SMEM(kernel, grid, block, smem, stream, ...)                                                  \
  do {                                                                                                               \
    static bool _attr_done = false;                                                                                  \
    if (!_attr_done && (smem) > 48 * 1024) {                                                                         \
      PG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(smem))); \
      _attr_done = true;                                                                                             \
    }                                                                                                                \
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                                              \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
