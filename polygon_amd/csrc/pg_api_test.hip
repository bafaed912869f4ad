// C-ABI entry points that expose single kernels for unit tests (tests/test_gpu_kernels.py).
#include "../../include/polygon_hip.h"
#include "pg_gemm.h"
#include "pg_scatter.h"

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

// pg_gemm_tn with the options the training step uses: fused column sums and a slab buffer for the two-stage split-K
// reduction (large bf16 outputs then take the 256 x 256 LDS-DMA kernel)
extern "C" int pg_gemm_tn_ws(int dtype, const void* A, long lda, const void* B, long ldb, int M, int N, int R, float* C,
                             long ldc, float alpha, float* colsum_a, float* colsum_b, float* slab, size_t slab_bytes,
                             void* stream) {
  if (dtype == PG_F32)
    return launch_gemm_tn<float>((const float*)A, lda, (const float*)B, ldb, M, N, R, C, ldc, alpha, nullptr, (hipStream_t)stream,
                                 colsum_a, colsum_b, slab, slab_bytes);
  if (dtype == PG_BF16)
    return launch_gemm_tn<bf16_t>((const bf16_t*)A, lda, (const bf16_t*)B, ldb, M, N, R, C, ldc, alpha, nullptr,
                                  (hipStream_t)stream, colsum_a, colsum_b, slab, slab_bytes);
  PG_FAIL(PG_ERR_ARG, "pg_gemm_tn_ws: bad dtype %d", dtype);
}

// ---- step-kernel micro-benchmark hook (development / roofline measurement) --------------------------
#include <stdlib.h>
#include "pg_gru.h"
template <typename T>
static int bench_gru_t(int which, int rows, int H, void* ws, size_t ws_bytes, int repeat, hipStream_t st) {
  const int dbg = getenv("PG_DBG") ? atoi(getenv("PG_DBG")) : 0;
  char* p = static_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = p + off; off = (off + bytes + 255) / 256 * 256; return r; };
  const size_t rh = (size_t)rows * H;
  T* hop0 = (T*)take(rh * sizeof(T)); T* hop1 = (T*)take(rh * sizeof(T));
  float* h32 = (float*)take(rh * 4);
  T* W = (T*)take((size_t)3 * H * H * sizeof(T));
  float* bhn = (float*)take((size_t)3 * H * 4);
  float* gi = (float*)take(rh * 3 * 4);
  T* sv[5]; for (int i = 0; i < 5; ++i) sv[i] = (T*)take(rh * sizeof(T));
  T* dGI = (T*)take(rh * 3 * sizeof(T)); T* dGH = (T*)take(rh * 3 * sizeof(T)); T* dGHin = (T*)take(rh * 3 * sizeof(T));
  float* dhz0 = (float*)take(rh * 4); float* dhz1 = (float*)take(rh * 4); float* dout = (float*)take(rh * 4);
  PG_REQUIRE(off <= ws_bytes, "bench_gru: workspace too small (%zu < %zu)", ws_bytes, off);
  for (int i = 0; i < repeat; ++i) {
    if (which == 0) {
      GruFwdArgs<T> a;
      a.rows = rows; a.H = H; a.hprev_op = (i & 1) ? hop1 : hop0; a.hprev32 = h32; a.Whh = W; a.bhn = bhn;
      a.gi_mode = PG_GI_ARRAY; a.gi = gi; a.h_op = (i & 1) ? hop0 : hop1; a.h32 = h32;
      a.sv_r = sv[0]; a.sv_z = sv[1]; a.sv_n = sv[2]; a.sv_hn = sv[3]; a.sv_hprev = sv[4];
      a.dbg = dbg;
      int s = launch_gru_fwd<T>(a, st);
      if (s) return s;
    } else {
      GruBwdArgs<T> a;
      a.rows_out = rows; a.rows_in = rows; a.H = H; a.dGH_in = dGHin; a.WhhT = W; a.dhz_in = dhz0; a.dout = dout;
      a.sv_r = sv[0]; a.sv_z = sv[1]; a.sv_n = sv[2]; a.sv_hn = sv[3]; a.sv_hprev = sv[4];
      a.dGI = dGI; a.dGH = dGH; a.dhz_out = dhz1; a.dbg = dbg;
      int s = launch_gru_bwd<T>(a, st);
      if (s) return s;
    }
  }
  return PG_OK;
}
extern "C" int pg_bench_gru(int dtype, int which, int rows, int H, void* ws, size_t ws_bytes, int repeat, void* stream) {
  if (dtype == PG_BF16) return bench_gru_t<bf16_t>(which, rows, H, ws, ws_bytes, repeat, (hipStream_t)stream);
  return bench_gru_t<float>(which, rows, H, ws, ws_bytes, repeat, (hipStream_t)stream);
}

// the embedding-table gradient (pg_scatter.h): sort by token + per-chunk sums + reduce, in one or several passes
extern "C" size_t pg_tok_scatter_scratch_bytes(long N, int C, int Vp) {
  if (N < 1 || C < 1 || Vp < 1 || Vp > PG_TS_MAX_VP) return 0;
  return tok_scatter_layout(std::min(N, tok_sort_max_rows(Vp)), C, Vp).total;
}
extern "C" int pg_tok_scatter_sum(int dtype, const int32_t* tok, const void* G, long ldg, long N, int C, int Vp, float* dT,
                                  float* colsum, void* scratch, size_t scratch_bytes, void* stream) {
  if (dtype == PG_F32)
    return launch_tok_scatter_sum<float>(tok, (const float*)G, ldg, N, C, Vp, dT, colsum, scratch, scratch_bytes, (hipStream_t)stream);
  if (dtype == PG_BF16)
    return launch_tok_scatter_sum<bf16_t>(tok, (const bf16_t*)G, ldg, N, C, Vp, dT, colsum, scratch, scratch_bytes, (hipStream_t)stream);
  PG_FAIL(PG_ERR_ARG, "pg_tok_scatter_sum: bad dtype %d", dtype);
}
