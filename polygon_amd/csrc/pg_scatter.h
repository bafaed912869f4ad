// Gradient of a gather table: dT[v, :] = sum of G[r, :] over the packed rows r with tok[r] = v (x_emb folded into the layer-0
// input projection, pg_misc.h gi_table_kernel; reference: the nn.Embedding + nn.GRU input of vae_model.py:70-96).
//
// A scatter-sum, HBM-bound (G is read once), not a GEMM.  What was measured on tools/scatter_bench.hip's shape (22 k rows x
// 1536 columns bf16, 64-row table, SMILES-like token skew; the decoder of the bench step has ~49 k rows):
//   one-hot^T . G on the MFMA engine (rounds 1-3)                       83 us in the step (half-empty 128-row tiles, a one-hot
//                                                                       operand per column tile)
//   LDS table per block, ds_add_f32                                     180 us: LDS float atomics retire ~1 lane per 3 clocks
//   LDS table per wave, plain read-modify-write (lanes own columns)     56 us: one LDS round trip per row, rows in sequence
//   rows sorted by token first, sums in registers (this file)           see DESIGN.md
// So: (1) tok_sort_kernel - one block - stable counting sort of the row indices by token, and a table of CHUNKS: runs of at most
// PG_TS_ROWS sorted rows of ONE token; (2) tok_chunk_sum_kernel - a block per (128-column group, chunk) - gathers the rows
// of its chunk with independent 16-byte loads, adds them in registers, folds the 16 row lanes through LDS and stores ONE
// partial row per chunk (plain stores); (3) tok_chunk_reduce_kernel - dT[v] = sum of v's chunks, and the sum of all chunks =
// the column sum of G (the bias gradient), which the GEMM used to fuse.  No atomics on the data path, nothing to zero.
#pragma once
#include "pg_common.h"
#include <algorithm>

#ifndef PG_TRY
#define PG_TRY(expr)            \
  do {                          \
    int _s = (expr);            \
    if (_s != PG_OK) return _s; \
  } while (0)
#endif

#ifndef PG_TS_ROWS
#define PG_TS_ROWS 128          // rows per chunk = 16 row lanes x 8 loads in flight per lane (a multiple of 128)
#endif
#define PG_TS_MAX_VP 128

struct TokScatterWs {
  size_t perm, chunks, tokchunk, slab, total;   // byte offsets into the scratch buffer
  int max_chunks;
};
// sum over tokens of ceil(count / PG_TS_ROWS) <= N / PG_TS_ROWS + (tokens present)
static inline TokScatterWs tok_scatter_layout(long N, int C, int Vp) {
  TokScatterWs w;
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t r = p; p = (p + bytes + 255) / 256 * 256; return r; };
  w.max_chunks = (int)(N / PG_TS_ROWS) + Vp;   // = tok_sort_chunks(N, Vp)
  w.perm = take((size_t)N * 4);
  w.chunks = take((size_t)w.max_chunks * sizeof(int4));
  w.tokchunk = take((size_t)(Vp + 1) * 4);
  w.slab = take((size_t)w.max_chunks * C * sizeof(float));
  w.total = p;
  return w;
}

// perm: row indices grouped by token, in a FIXED order within a token (the sums downstream are deterministic); chunks[k] =
// {token, first, last + 1 (positions in perm), 0}; tokchunk[v] = first chunk of token v, tokchunk[Vp] = number of chunks.
// Rows whose token is outside [0, Vp) are left out.
// ONE block, no atomics.  (Measured, 22 k rows: atomicAdd on LDS counters 25 us - LDS atomics of any type retire about one
// lane per three clocks and the block's 16 waves share the one LDS; counters in registers with a ballot per distinct token
// among 64 rows 100 us - ~14 scalar round trips per 64 rows.)  Here every THREAD owns the rows t, t + NT, t + 2 NT, ... and a
// private BYTE counter per token, cnt8[v][t] (at most 255 rows per thread and launch: the launcher splits longer inputs):
// plain LDS read-modify-write, all threads in parallel.  The position of thread t's k-th row of token v is then
//   start[v] + (rows of v in threads < t) + k,
// the middle term = base[v][t / 16] (a wave scan over the 16-thread group totals, 4 v_sad_u8 per group) + the bytes of the
// threads in front of t in its own group (one 16-byte LDS read, masked, 4 v_sad_u8), and k a second private byte counter.
// (token stride NT + 16 bytes: the counters of different tokens at the same thread fall into different LDS banks)
#define PG_TS_SORT_LDS(Vp, NT) ((size_t)2 * (Vp) * ((NT) + 16) + (size_t)(Vp) * ((NT) / 16) * 4 + (size_t)(3 * (Vp) + 1) * 4)
__device__ __forceinline__ unsigned pg_bytesum(unsigned x, unsigned acc) { return __builtin_amdgcn_sad_u8(x, 0u, acc); }
template <int NT>
static __global__ __launch_bounds__(NT) void tok_sort_kernel(const int* __restrict__ tok, int N, int Vp, int* __restrict__ perm,
                                                             int4* __restrict__ chunks, int* __restrict__ tokchunk) {
  constexpr int NW = NT / 64, NG = NT / 16, LD = NT + 16;
  extern __shared__ __align__(16) unsigned char ts_smem[];
  unsigned char* cnt8 = ts_smem;                      // [Vp][LD]
  unsigned char* run8 = ts_smem + (size_t)Vp * LD;    // [Vp][LD]
  int* base = reinterpret_cast<int*>(ts_smem + (size_t)2 * Vp * LD);   // [Vp][NG]
  int* start = base + Vp * NG;                        // [Vp]
  int* cnt = start + Vp;                              // [Vp]
  int* cstart = cnt + Vp;                             // [Vp + 1]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int i = t; i < 2 * Vp * LD / 4; i += NT) reinterpret_cast<unsigned*>(ts_smem)[i] = 0u;
  __syncthreads();
  constexpr int U = 8;                                // tokens fetched per round: the loads of a round are in flight together
  for (int i0 = t; i0 < N; i0 += U * NT) {
    int v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = i0 + u * NT < N ? tok[i0 + u * NT] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((unsigned)v[u] < (unsigned)Vp) cnt8[v[u] * LD + t] += 1;
  }
  __syncthreads();
  for (int v = w; v < Vp; v += NW) {                  // (uniform per wave)
    int tot = 0;
    if (lane < NG) {
      const uint4 q = *reinterpret_cast<const uint4*>(cnt8 + v * LD + lane * 16);
      tot = (int)pg_bytesum(q.w, pg_bytesum(q.z, pg_bytesum(q.y, pg_bytesum(q.x, 0u))));
    }
    int inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane < NG) base[v * NG + lane] = inc - tot;
    if (lane == 63) cnt[v] = inc;
  }
  __syncthreads();
  if (w == 0) {                                       // exclusive scans over the tokens: rows -> start, chunks -> cstart
    const int a0 = lane < Vp ? cnt[lane] : 0, a1 = 64 + lane < Vp ? cnt[64 + lane] : 0;
    const int b0 = (a0 + PG_TS_ROWS - 1) / PG_TS_ROWS, b1 = (a1 + PG_TS_ROWS - 1) / PG_TS_ROWS;
    int ra0 = a0, ra1 = a1, rb0 = b0, rb1 = b1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int oa0 = __shfl_up(ra0, d), oa1 = __shfl_up(ra1, d), ob0 = __shfl_up(rb0, d), ob1 = __shfl_up(rb1, d);
      if (lane >= d) { ra0 += oa0; ra1 += oa1; rb0 += ob0; rb1 += ob1; }
    }
    const int ta = __shfl(ra0, 63), tb = __shfl(rb0, 63);          // totals of tokens 0..63
    if (lane < Vp) { start[lane] = ra0 - a0; cstart[lane] = rb0 - b0; }
    if (64 + lane < Vp) { start[64 + lane] = ta + ra1 - a1; cstart[64 + lane] = tb + rb1 - b1; }
    if (lane == 63) cstart[Vp] = tb + rb1;
  }
  __syncthreads();
  if (t <= Vp) tokchunk[t] = cstart[t];
  if (t < Vp) {
    const int n = cstart[t + 1] - cstart[t], last = start[t] + cnt[t];
    for (int k = 0; k < n; ++k)
      chunks[cstart[t] + k] = make_int4(t, start[t] + k * PG_TS_ROWS, min(last, start[t] + (k + 1) * PG_TS_ROWS), 0);
  }
  const int keep = t & 15;                            // bytes of my group that lie in front of me
  unsigned mask[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kb = min(4, max(0, keep - 4 * j));
    mask[j] = kb >= 4 ? 0xffffffffu : ((1u << (8 * kb)) - 1u);
  }
  for (int i0 = t; i0 < N; i0 += U * NT) {
    int vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) vv[u] = i0 + u * NT < N ? tok[i0 + u * NT] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int v = vv[u];
      if ((unsigned)v >= (unsigned)Vp) continue;
      const uint4 q = *reinterpret_cast<const uint4*>(cnt8 + v * LD + (t & ~15));
      const int front = (int)pg_bytesum(q.w & mask[3], pg_bytesum(q.z & mask[2], pg_bytesum(q.y & mask[1], pg_bytesum(q.x & mask[0], 0u))));
      const int k = run8[v * LD + t];
      run8[v * LD + t] = (unsigned char)(k + 1);
      perm[start[v] + base[v * NG + (t >> 4)] + front + k] = i0 + u * NT;
    }
  }
}

// slab[chunk, c0 .. c0 + CC) = sum over the chunk's rows of G[row, c0 .. c0 + CC);  grid (column groups, max_chunks), 256 threads
template <typename T>
static __global__ __launch_bounds__(256) void tok_chunk_sum_kernel(const T* __restrict__ G, long ldg, int C, const int* __restrict__ perm,
                                                                   const int4* __restrict__ chunks, const int* __restrict__ tokchunk,
                                                                   int Vp, float* __restrict__ slab) {
  constexpr int EPC = 16 / (int)sizeof(T), CC = 16 * EPC, NL = PG_TS_ROWS / 16;
  __shared__ float red[16][CC + 1];
  const int chunk = blockIdx.y;
  if (chunk >= tokchunk[Vp]) return;                    // (uniform: the grid covers the bound, not the count)
  const int4 ch = chunks[chunk];
  const int l16 = threadIdx.x & 15, rsub = threadIdx.x >> 4;
  const int c0 = blockIdx.x * CC, col = c0 + l16 * EPC;
  float acc[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) acc[j] = 0.0f;
  if (col < C) {                                        // (C is a multiple of EPC: whole chunks of columns)
    uint4 q[NL];
    bool ok[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int i = ch.y + rsub + 16 * u;
      ok[u] = i < ch.z;
      if (ok[u]) q[u] = *reinterpret_cast<const uint4*>(G + (long)perm[i] * ldg + col);
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      if (!ok[u]) continue;
      const T* e = reinterpret_cast<const T*>(&q[u]);
#pragma unroll
      for (int j = 0; j < EPC; ++j) acc[j] += to_f32<T>(e[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < EPC; ++j) red[rsub][l16 * EPC + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < CC && c0 + threadIdx.x < C) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][threadIdx.x];
    slab[(long)chunk * C + c0 + threadIdx.x] = s;
  }
}

// dT[v, c] = sum of the chunks of token v (block row v < Vp; zero when the token does not occur); block row Vp: colsum[c] +=
// the sum of ALL chunks (optional).  grid (ceil(C / 32), Vp + 1), 256 threads = 32 columns x 8 chunk lanes: the common
// token's ~80 chunks and the column sum's ~240 are 10 / 30 loads per thread instead of a serial string.
static __global__ __launch_bounds__(256) void tok_chunk_reduce_kernel(const float* __restrict__ slab, const int* __restrict__ tokchunk,
                                                                      int Vp, int C, float* __restrict__ dT, float* __restrict__ colsum,
                                                                      int accumulate) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, v = blockIdx.y;
  if (v == Vp && !colsum) return;
  const int k0 = v < Vp ? tokchunk[v] : 0, k1 = v < Vp ? tokchunk[v + 1] : tokchunk[Vp];
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (c < C) {
    const float* p = slab + c;
    int k = k0 + kl;
    for (; k + 24 < k1; k += 32) {
      s0 += p[(long)k * C]; s1 += p[(long)(k + 8) * C]; s2 += p[(long)(k + 16) * C]; s3 += p[(long)(k + 24) * C];
    }
    for (; k < k1; k += 8) s0 += p[(long)k * C];
  }
  red[kl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kl == 0 && c < C) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][cl];
    if (v < Vp) dT[(long)v * C + c] = accumulate ? dT[(long)v * C + c] + s : s;
    else atomicAdd(colsum + c, s);                      // (one add per address: the buffer accumulates other gradients' bias terms)
  }
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------
// The sort needs the tokens only: the training forward pass runs it beside the encoder (side stream) into tables that stay
// in the workspace, and the backward pass finds the rows already grouped.
struct TokSort {
  int* perm = nullptr;       // [N]
  int4* chunks = nullptr;    // [tok_sort_chunks(N, Vp)]
  int* tokchunk = nullptr;   // [Vp + 1]
};
static inline int tok_sort_chunks(long N, int Vp) { return (int)(N / PG_TS_ROWS) + Vp; }
// rows one sort launch takes: byte counters, 255 rows per thread
static inline long tok_sort_max_rows(int Vp) { return (long)(255 * (Vp <= 64 ? 1024 : 512)) / PG_TS_ROWS * PG_TS_ROWS; }
static inline int launch_tok_sort(const int* tok, long N, int Vp, const TokSort& ts, hipStream_t st) {
  PG_REQUIRE(tok && ts.perm && ts.chunks && ts.tokchunk, "tok_sort: NULL buffer");
  PG_REQUIRE(N >= 1 && N <= tok_sort_max_rows(Vp) && Vp >= 1 && Vp <= PG_TS_MAX_VP, "tok_sort: N=%ld Vp=%d out of range", N, Vp);
  if (Vp <= 64) {
    PG_LAUNCH_BIG_SMEM(tok_sort_kernel<1024>, dim3(1), dim3(1024), PG_TS_SORT_LDS(Vp, 1024), st, tok, (int)N, Vp, ts.perm, ts.chunks,
                       ts.tokchunk);
  } else {
    PG_LAUNCH_BIG_SMEM(tok_sort_kernel<512>, dim3(1), dim3(512), PG_TS_SORT_LDS(Vp, 512), st, tok, (int)N, Vp, ts.perm, ts.chunks,
                       ts.tokchunk);
  }
  PG_CHECK_LAUNCH();
  return PG_OK;
}
// dT (+)= the per-token sums of G's rows, grouped by `ts` (a launch_tok_sort of the same N tokens); slab: tok_sort_chunks(N, Vp)
// rows of C floats
template <typename T>
static inline int launch_tok_sum_sorted(const T* G, long ldg, long N, int C, int Vp, const TokSort& ts, float* slab, size_t slab_bytes,
                                        float* dT, float* colsum, int accumulate, hipStream_t st) {
  constexpr int EPC = 16 / (int)sizeof(T), CC = 16 * EPC;
  PG_REQUIRE(G && dT && slab && ts.perm && ts.chunks && ts.tokchunk, "tok_sum_sorted: NULL buffer");
  PG_REQUIRE(C >= EPC && C % EPC == 0 && ldg % EPC == 0 && ldg >= C && ((uintptr_t)G % 16) == 0,
             "tok_sum_sorted: rows must be whole 16-byte chunks (C=%d ld=%ld)", C, ldg);
  PG_REQUIRE(N >= 1 && N <= tok_sort_max_rows(Vp) && Vp >= 1 && Vp <= PG_TS_MAX_VP, "tok_sum_sorted: N=%ld Vp=%d out of range", N, Vp);
  const int max_chunks = tok_sort_chunks(N, Vp);
  PG_REQUIRE(slab_bytes >= (size_t)max_chunks * C * sizeof(float), "tok_sum_sorted: slab too small (%zu < %zu)", slab_bytes,
             (size_t)max_chunks * C * sizeof(float));
  hipLaunchKernelGGL(tok_chunk_sum_kernel<T>, dim3((C + CC - 1) / CC, max_chunks), dim3(256), 0, st, G, ldg, C, ts.perm, ts.chunks,
                     ts.tokchunk, Vp, slab);
  PG_CHECK_LAUNCH();
  hipLaunchKernelGGL(tok_chunk_reduce_kernel, dim3((C + 31) / 32, Vp + 1), dim3(256), 0, st, slab, ts.tokchunk, Vp, C, dT, colsum,
                     accumulate);
  PG_CHECK_LAUNCH();
  return PG_OK;
}

// Sort and sum in one call.  scratch: tok_scatter_layout(N, C, Vp).total bytes for a single pass, 256-byte aligned, free for this
// stream's use until the call's kernels end; with less (or more rows than one sort takes) the rows go through in several passes.
template <typename T>
static inline int launch_tok_scatter_sum(const int* tok, const T* G, long ldg, long N, int C, int Vp, float* dT, float* colsum,
                                         void* scratch, size_t scratch_bytes, hipStream_t st) {
  PG_REQUIRE(tok && G && dT && scratch, "tok_scatter_sum: NULL buffer");
  PG_REQUIRE(N >= 1 && N < (1L << 30) && Vp >= 1 && Vp <= PG_TS_MAX_VP && C >= 1, "tok_scatter_sum: N=%ld Vp=%d out of range", N, Vp);
  PG_REQUIRE(((uintptr_t)scratch % 256) == 0 && scratch_bytes >= tok_scatter_layout(std::min<long>(N, PG_TS_ROWS), C, Vp).total,
             "tok_scatter_sum: scratch too small (%zu bytes)", scratch_bytes);
  long per_pass = std::min(N, tok_sort_max_rows(Vp));
  if (tok_scatter_layout(per_pass, C, Vp).total > scratch_bytes) {
    const size_t fixed = tok_scatter_layout(PG_TS_ROWS, C, Vp).total + 4096;          // the Vp chunks of slack, tables, padding
    const size_t per_chunk = (size_t)PG_TS_ROWS * 4 + sizeof(int4) + (size_t)C * sizeof(float);
    per_pass = std::min(per_pass, (long)((scratch_bytes - std::min(scratch_bytes, fixed)) / per_chunk) * PG_TS_ROWS);
    while (per_pass > PG_TS_ROWS && tok_scatter_layout(per_pass, C, Vp).total > scratch_bytes) per_pass -= PG_TS_ROWS;
    PG_REQUIRE(per_pass >= PG_TS_ROWS, "tok_scatter_sum: scratch too small (%zu bytes)", scratch_bytes);
  }
  for (long n0 = 0; n0 < N; n0 += per_pass) {
    const long n = std::min(per_pass, N - n0);
    const TokScatterWs w = tok_scatter_layout(n, C, Vp);
    PG_REQUIRE(w.total <= scratch_bytes, "tok_scatter_sum: internal: pass of %ld rows needs %zu > %zu bytes", n, w.total, scratch_bytes);
    char* base = static_cast<char*>(scratch);
    TokSort ts;
    ts.perm = reinterpret_cast<int*>(base + w.perm);
    ts.chunks = reinterpret_cast<int4*>(base + w.chunks);
    ts.tokchunk = reinterpret_cast<int*>(base + w.tokchunk);
    PG_TRY(launch_tok_sort(tok + n0, n, Vp, ts, st));
    PG_TRY(launch_tok_sum_sorted<T>(G + n0 * ldg, ldg, n, C, Vp, ts, reinterpret_cast<float*>(base + w.slab),
                                    (size_t)w.max_chunks * C * sizeof(float), dT, colsum, n0 > 0 ? 1 : 0, st));
  }
  return PG_OK;
}
