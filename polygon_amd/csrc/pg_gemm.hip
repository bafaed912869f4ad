// Batched GEMM kernels (NT and TN forms) on the gfx950 MFMA tile engines.
#include "pg_gemm.h"
#include "pg_mma.h"

template <class Cfg, bool DROP>
__global__ __launch_bounds__(Cfg::NTHREADS) void gemm_nt_kernel(const typename Cfg::T* __restrict__ A, long lda,
                                                                 const typename Cfg::T* __restrict__ B, long ldb, int M,
                                                                 int N, int K, GemmEpi<typename Cfg::T> epi) {
  using T = typename Cfg::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (N + Cfg::BN - 1) / Cfg::BN;
  const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
  const int m0 = bm * Cfg::BM, n0 = bn * Cfg::BN;
  f32x16 acc[Cfg::TM][Cfg::TN];
{
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = zacc;
  }
  RowSrc<T> as{A, lda, m0, M};
  RowSrc<T> bs{B, ldb, n0, N};
  mma_mainloop_nt<Cfg, Cfg::TN, Cfg::NGATE - 1>(acc, as, bs, K, smem);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  float alpha = epi.alpha;
  if (epi.alpha_dev) alpha *= *epi.alpha_dev;
  const float inv_keep = DROP ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
#pragma unroll
  for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + Cfg::row_of(wm, tm, reg, lane);
      const bool mok = m < M;
      int rt = 0, rb = 0;
      if (DROP && mok) {
        rt = epi.row_t[m];
        rb = epi.row_b[m] * epi.b_mul + epi.b_add;
      }
#pragma unroll
      for (int tn = 0; tn < Cfg::TN; ++tn) {
        const int n = n0 + Cfg::col_in_group(wn, tn, lane & 31);
        float v = acc[tm][tn][reg] * alpha;
        if (!mok || n >= N) continue;
        if (epi.bias) v += epi.bias[n];
        if (DROP) v *= dropout_scale(epi.seed, epi.layer, rt, rb, n, epi.p_drop, inv_keep);
        const long o = (long)m * epi.ldc + n;
        if (epi.C32) {
          if (epi.accumulate) v += epi.C32[o];
          epi.C32[o] = v;
        }
        if (epi.Ct) epi.Ct[o] = from_f32<T>(v);
      }
    }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::NTHREADS) void gemm_tn_kernel(const typename Cfg::T* __restrict__ A, long lda,
                                                                 const typename Cfg::T* __restrict__ B, long ldb, int M,
                                                                 int N, int R, int rows_per_split, float* C, long ldc,
                                                                 float alpha, const float* alpha_dev) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (N + Cfg::BN - 1) / Cfg::BN;
  const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
  const int m0 = bm * Cfg::BM, n0 = bn * Cfg::BN;
  const int r0 = blockIdx.y * rows_per_split;
  const int r1 = min(R, r0 + rows_per_split);
  f32x16 acc[Cfg::TM][Cfg::TN];
{
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = zacc;
  }
  mma_mainloop_tn<Cfg>(acc, A, lda, m0, M, B, ldb, n0, N, r0, r1, smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  if (alpha_dev) alpha *= *alpha_dev;
#pragma unroll
  for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + Cfg::row_of(wm, tm, reg, lane);
#pragma unroll
      for (int tn = 0; tn < Cfg::TN; ++tn) {
        const int n = n0 + Cfg::col_of(wn, tn, lane);
        const float v = acc[tm][tn][reg] * alpha;
        if (m < M && n < N) atomicAdd(C + (long)m * ldc + n, v);
      }
    }
}

template <typename T>
int launch_gemm_nt(const T* A, long lda, const T* B, long ldb, int M, int N, int K, const GemmEpi<T>& epi,
                   hipStream_t stream) {
  constexpr int EPC = 16 / sizeof(T);
  if (M <= 0 || N <= 0) return PG_OK;
  PG_REQUIRE(K % EPC == 0 && lda % EPC == 0 && ldb % EPC == 0, "gemm_nt: K/lda/ldb (%d,%ld,%ld) must be multiples of %d",
             K, lda, ldb, EPC);
  PG_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_nt: operands must be 16-byte aligned");
  PG_REQUIRE(!epi.accumulate || epi.C32, "gemm_nt: accumulate needs an fp32 output");
  if (N <= 64) {
    using Cfg = TileCfg<T, 128, 64, 8, 4, 1>;
    dim3 grid(cdiv(M, Cfg::BM) * cdiv(N, Cfg::BN));
    if (epi.p_drop > 0.0f)
      PG_LAUNCH_BIG_SMEM((gemm_nt_kernel<Cfg, true>), grid, dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, A, lda, B, ldb, M, N, K, epi);
    else
      PG_LAUNCH_BIG_SMEM((gemm_nt_kernel<Cfg, false>), grid, dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, A, lda, B, ldb, M, N, K, epi);
  } else {
    using Cfg = TileCfg<T, 128, 128, 8, 2, 2>;
    dim3 grid(cdiv(M, Cfg::BM) * cdiv(N, Cfg::BN));
    if (epi.p_drop > 0.0f)
      PG_LAUNCH_BIG_SMEM((gemm_nt_kernel<Cfg, true>), grid, dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, A, lda, B, ldb, M, N, K, epi);
    else
      PG_LAUNCH_BIG_SMEM((gemm_nt_kernel<Cfg, false>), grid, dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, A, lda, B, ldb, M, N, K, epi);
  }
  PG_CHECK_LAUNCH();
  return PG_OK;
}

template <typename T>
int launch_gemm_tn(const T* A, long lda, const T* B, long ldb, int M, int N, int R, float* C, long ldc, float alpha,
                   const float* alpha_dev, hipStream_t stream) {
  constexpr int EPC = 16 / sizeof(T);
  if (M <= 0 || N <= 0 || R <= 0) return PG_OK;
  PG_REQUIRE(lda % EPC == 0 && ldb % EPC == 0, "gemm_tn: lda/ldb must be multiples of %d", EPC);
  PG_REQUIRE(lda >= (M + EPC - 1) / EPC * EPC && ldb >= (N + EPC - 1) / EPC * EPC, "gemm_tn: rows must be padded to whole chunks");
  PG_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_tn: operands must be 16-byte aligned");
  using Cfg = TileTN<T, 128, 128, 32, 2, 2>;
  const int tiles = cdiv(M, Cfg::BM) * cdiv(N, Cfg::BN);
  int splits = max(1, min(cdiv(R, Cfg::BKR * 4), cdiv(1024, tiles)));
  int rows_per_split = cdiv(cdiv(R, splits), Cfg::BKR) * Cfg::BKR;
  splits = cdiv(R, rows_per_split);
  dim3 grid(tiles, splits);
  PG_LAUNCH_BIG_SMEM(gemm_tn_kernel<Cfg>, grid, dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, A, lda, B, ldb, M, N, R,
                     rows_per_split, C, ldc, alpha, alpha_dev);
  PG_CHECK_LAUNCH();
  return PG_OK;
}

template int launch_gemm_nt<float>(const float*, long, const float*, long, int, int, int, const GemmEpi<float>&, hipStream_t);
template int launch_gemm_nt<bf16_t>(const bf16_t*, long, const bf16_t*, long, int, int, int, const GemmEpi<bf16_t>&, hipStream_t);
template int launch_gemm_tn<float>(const float*, long, const float*, long, int, int, int, float*, long, float, const float*, hipStream_t);
template int launch_gemm_tn<bf16_t>(const bf16_t*, long, const bf16_t*, long, int, int, int, float*, long, float, const float*, hipStream_t);
