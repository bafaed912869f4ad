// Host-side helpers shared by the entry-point files (pg_vae.hip, pg_encgen.hip): error propagation, workspace addressing,
// launchers of the small kernels of pg_misc.h.
#pragma once
#include <algorithm>
#include "pg_misc.h"

#define PG_TRY(expr)            \
  do {                          \
    int _s = (expr);            \
    if (_s != PG_OK) return _s; \
  } while (0)

template <typename T> static inline T* at(void* base, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }
template <typename T> static inline const T* at(const void* base, size_t off) {
  return reinterpret_cast<const T*>(static_cast<const char*>(base) + off);
}
static inline int ew_blocks(long n) { return (int)std::min<long>(cdiv(n, 256), 4096); }

template <typename T> static int convert_to(const float* in, T* out, long n, hipStream_t st) {
  hipLaunchKernelGGL(convert_kernel<T>, dim3(ew_blocks(n)), dim3(256), 0, st, in, out, n);
  PG_CHECK_LAUNCH();
  return PG_OK;
}
template <typename T>
static int colsum(const T* X, long R, int C, long ld, float* out, float alpha, const float* alpha_dev, hipStream_t st) {
  if (R <= 0) return PG_OK;
  const int ysplit = (int)std::max<long>(1, std::min<long>(cdiv(R, 64), 128));
  const long rpb = cdiv(R, ysplit);
  hipLaunchKernelGGL(colsum_kernel<T>, dim3(cdiv(C, 64), cdiv(R, rpb)), dim3(256), 0, st, X, R, C, ld, rpb, out, alpha, alpha_dev);
  PG_CHECK_LAUNCH();
  return PG_OK;
}
static int small_gemm(const float* A, long sam, long sak, const float* Bm, long sbk, long sbn, int M, int N, int K, float* C,
                      long ldc, int skip_row, hipStream_t st) {
  hipLaunchKernelGGL(small_gemm_kernel, dim3(cdiv(N, 16), cdiv(M, 16), cdiv(K, PG_SG_KC)), dim3(256), 0, st, A, sam, sak, Bm, sbk,
                     sbn, M, N, K, C, ldc, skip_row);
  PG_CHECK_LAUNCH();
  return PG_OK;
}

int check_persist();   // pg_vae.hip: sticky give-up word of the persistent kernels -> PG_ERR_* at entry

// pg_heads.hip: the row-local stretches between the recurrences as one kernel each (bf16).  Return false when the shape is
// not covered (nothing was launched: the caller takes the separate GEMM + element-wise launches); *status = launch result.
bool launch_enc_heads(const bf16_t* hfin, int B, int K, const bf16_t* wq, const float* bq, int Dz, const float* eps, float* mulv,
                      float* z, bf16_t* z_op, float* mu_out, float* lv_out, float* eps_save, float* kl_sum, float inv_B,
                      float* kl_out, hipStream_t st, int* status);
bool launch_dec_zin(const float* z, int B, int Dz, int Hd, const bf16_t* wih0z, const bf16_t* latw, const float* latb, bf16_t* z_op,
                    float* zproj, float* h0_32, bf16_t* h0_op, hipStream_t st, int* status);
