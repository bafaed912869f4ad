// C-ABI entry points that expose single kernels for unit tests (tests/test_gpu_kernels.py).
#include "../../include/polygon_hip.h"
#include "pg_gemm.h"

template <typename T>
static int gemm_nt_t(const void* A, long lda, const void* B, long ldb, int M, int N, int K, const float* bias,
                     float* C32, void* Ct, long ldc, int accumulate, float alpha, hipStream_t s) {
  GemmEpi<T> e;
  e.bias = bias; e.C32 = C32; e.Ct = (T*)Ct; e.ldc = ldc; e.accumulate = accumulate; e.alpha = alpha;
  return launch_gemm_nt<T>((const T*)A, lda, (const T*)B, ldb, M, N, K, e, s);
}

extern "C" int pg_gemm_nt(int dtype, const void* A, long lda, const void* B, long ldb, int M, int N, int K,
                          const float* bias, float* C32, void* Ct, long ldc, int accumulate, float alpha, void* stream) {
  if (dtype == PG_F32) return gemm_nt_t<float>(A, lda, B, ldb, M, N, K, bias, C32, Ct, ldc, accumulate, alpha, (hipStream_t)stream);
  if (dtype == PG_BF16) return gemm_nt_t<bf16_t>(A, lda, B, ldb, M, N, K, bias, C32, Ct, ldc, accumulate, alpha, (hipStream_t)stream);
  PG_FAIL(PG_ERR_ARG, "pg_gemm_nt: bad dtype %d", dtype);
}

extern "C" int pg_gemm_tn(int dtype, const void* A, long lda, const void* B, long ldb, int M, int N, int R, float* C,
                          long ldc, float alpha, void* stream) {
  if (dtype == PG_F32) return launch_gemm_tn<float>((const float*)A, lda, (const float*)B, ldb, M, N, R, C, ldc, alpha, nullptr, (hipStream_t)stream);
  if (dtype == PG_BF16) return launch_gemm_tn<bf16_t>((const bf16_t*)A, lda, (const bf16_t*)B, ldb, M, N, R, C, ldc, alpha, nullptr, (hipStream_t)stream);
  PG_FAIL(PG_ERR_ARG, "pg_gemm_tn: bad dtype %d", dtype);
}
