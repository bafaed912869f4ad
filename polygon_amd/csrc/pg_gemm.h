// Batched (non-recurrent) GEMM launchers built on the tile engines of pg_mma.h.
#pragma once
#include "pg_common.h"

// Epilogue options of gemm_nt: C = alpha * A.B^T (+ bias[n]) (+ C32 if accumulate), optionally multiplied by the
// inter-layer dropout mask of element (layer, row_t[m], row_b[m], n) (used by the dX GEMMs of the decoder backward).
template <typename T> struct GemmEpi {
  const float* bias = nullptr;       // [N]
  float* C32 = nullptr;              // fp32 output (nullable)
  T* Ct = nullptr;                   // operand-dtype output (nullable)
  long ldc = 0;
  int accumulate = 0;                // C32 += (needs C32)
  float alpha = 1.0f;
  const float* alpha_dev = nullptr;  // optional device scalar multiplied into alpha
  // dropout mask (p_drop > 0 enables)
  float p_drop = 0.0f;
  unsigned long long seed = 0;
  int layer = 0;
  const int* row_t = nullptr;
  const int* row_b = nullptr;
  int b_mul = 1, b_add = 0;
};

// C[M,N] = A[M,K] . B[N,K]^T ; lda/ldb in elements, K % (16/sizeof(T)) == 0, 16-byte aligned rows.
template <typename T>
int launch_gemm_nt(const T* A, long lda, const T* B, long ldb, int M, int N, int K, const GemmEpi<T>& epi,
                   hipStream_t stream);

// C[M,N] (fp32, ldc) += alpha * A[R,M]^T . B[R,N]  (atomic accumulation, C must be initialised by the caller).
// colsum_a[M] / colsum_b[N] (optional): += alpha * column sums of A / B over the R rows (bias gradients), taken from the
// chunks the kernel stages anyway - no second pass over the operand.
template <typename T>
int launch_gemm_tn(const T* A, long lda, const T* B, long ldb, int M, int N, int R, float* C, long ldc, float alpha,
                   const float* alpha_dev, hipStream_t stream, float* colsum_a = nullptr, float* colsum_b = nullptr,
                   float* partial = nullptr, size_t partial_bytes = 0);
// partial (optional, partial_bytes): scratch for a two-stage split-K reduction - every K split stores its tiles there with
// plain stores and a second kernel sums them into C - used when it is large enough (PG_TN_SLAB_BYTES covers every shape
// the launcher produces) and the output has enough tiles for the atomics to matter; C is then read-modify-written by the
// reduction kernel in stream order, so no other stream may add to the same C region meanwhile.
constexpr size_t PG_TN_SLAB_BYTES = (size_t)776 * 128 * 128 * sizeof(float);   // 768 split-K blocks, rounded up to whole XCD rounds

// C[M,N] (fp32, ldc) += alpha * A[M,K] . B[N,K]^T, K split over blocks (atomic accumulation into an initialised C): for
// outputs of a few tiles with a long K, where one block per tile is a latency chain.
template <typename T>
int launch_gemm_nt_splitk(const T* A, long lda, const T* B, long ldb, int M, int N, int K, float* C, long ldc, float alpha,
                          hipStream_t stream);

// workgroups the next gemm_tn256 launches aim for (0: the default, 192); see decoder_bwd_t
extern int g_tn256_blocks_hint;
