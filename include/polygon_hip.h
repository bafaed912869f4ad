/* polygon_hip.h -- C ABI of libpolygon_hip.so (MI355X / gfx950 only).
 *
 * Every entry point returns 0 on success or a negative PG_ERR_* code; pg_last_error() gives the
 * message of the calling thread's last failure.  All device pointers are raw HIP device pointers,
 * all work is enqueued on the caller's stream (a hipStream_t passed as void*), nothing is
 * allocated by the library and no C++ exception crosses the boundary.
 */
#ifndef POLYGON_HIP_H
#define POLYGON_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PG_API __attribute__((visibility("default")))
#define PG_VERSION 1
#define PG_F32 0
#define PG_BF16 1

PG_API int pg_version(void);
PG_API const char* pg_last_error(void);

/* Kernel-level test hooks (no reference counterpart; they expose the MFMA tile engines that every
 * entry point below is built from, so tests can check them in isolation).
 * pg_gemm_nt: C[M,N] = alpha * A[M,K] . B[N,K]^T (+bias[N]) (+C32 if accumulate) -> C32 (fp32) and/or Ct (dtype).
 * pg_gemm_tn: C[M,N] += alpha * A[R,M]^T . B[R,N]  (fp32 atomics into C). */
PG_API int pg_gemm_nt(int dtype, const void* A, long lda, const void* B, long ldb, int M, int N, int K, const float* bias,
               float* C32, void* Ct, long ldc, int accumulate, float alpha, void* stream);
PG_API int pg_gemm_tn(int dtype, const void* A, long lda, const void* B, long ldb, int M, int N, int R, float* C, long ldc,
               float alpha, void* stream);

#ifdef __cplusplus
}
#endif
#endif
