// Batched GEMM kernels (NT and TN forms) on the gfx950 MFMA tile engines.
#include <stdlib.h>
#include <type_traits>
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

// Split-K form for skinny outputs (few tiles, long K: the dz GEMMs behind the decoder BPTT, 8 tiles x K = 1536): one block
// per (tile, K split), fp32 atomics into an initialised C.  8 blocks walking 24 K tiles each are a 38-us latency chain
// on the step's critical path; 64 blocks of 3 tiles take a fifth of that.
template <class Cfg>
__global__ __launch_bounds__(Cfg::NTHREADS) void gemm_nt_splitk_kernel(const typename Cfg::T* __restrict__ A, long lda,
                                                                        const typename Cfg::T* __restrict__ B, long ldb, int M,
                                                                        int N, int K, int kt_per_split, float* __restrict__ C,
                                                                        long ldc, float alpha) {
  using T = typename Cfg::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (N + Cfg::BN - 1) / Cfg::BN, tiles = ((M + Cfg::BM - 1) / Cfg::BM) * tiles_n;
  const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
  const int bm = tile / tiles_n, bn = tile % tiles_n;
  const int m0 = bm * Cfg::BM, n0 = bn * Cfg::BN;
  const int nkt = (K + Cfg::BK - 1) / Cfg::BK;
  const int kt0 = split * kt_per_split;
  const int nk = min(kt_per_split, nkt - kt0);
  if (nk <= 0) return;
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
  mma_mainloop_nt<Cfg, Cfg::TN, Cfg::NGATE - 1>(acc, as, bs, K, smem, (int)threadIdx.x, kt0, nk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
#pragma unroll
  for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + Cfg::row_of(wm, tm, reg, lane);
#pragma unroll
      for (int tn = 0; tn < Cfg::TN; ++tn) {
        const int n = n0 + Cfg::col_in_group(wn, tn, lane & 31);
        if (m < M && n < N) atomicAdd(C + (long)m * ldc + n, acc[tm][tn][reg] * alpha);
      }
    }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::NTHREADS) void gemm_tn_kernel(const typename Cfg::T* __restrict__ A, long lda,
                                                                 const typename Cfg::T* __restrict__ B, long ldb, int M,
                                                                 int N, int R, int rows_per_split, int nsplits, float* C,
                                                                 long ldc, float alpha, const float* alpha_dev,
                                                                 float* colsum_a, float* colsum_b, float* partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware map: workgroups go round-robin over the 8 XCDs (private L2 each), so every tile of one row split is placed
  // on the SAME XCD - the split's rows of A and B then cross the fabric once instead of once per XCD that holds a tile.
  const int tiles_n = (N + Cfg::BN - 1) / Cfg::BN, tiles = ((M + Cfg::BM - 1) / Cfg::BM) * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int split = (slot / tiles) * 8 + xcd, tile = slot % tiles;
  if (split >= nsplits) return;
  const int bm = tile / tiles_n, bn = tile % tiles_n;
  const int m0 = bm * Cfg::BM, n0 = bn * Cfg::BN;
  const int r0 = split * rows_per_split;
  const int r1 = min(R, r0 + rows_per_split);
  f32x16 acc[Cfg::TM][Cfg::TN];
  {
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = zacc;
  }
  // fused column sums (bias gradients): taken by the tiles that sit in the first tile column of the OTHER operand
  const bool csa = colsum_a != nullptr && bn == 0, csb = colsum_b != nullptr && bm == 0;
  float ca[Cfg::EPC], cb[Cfg::EPC];
#pragma unroll
  for (int e = 0; e < Cfg::EPC; ++e) ca[e] = cb[e] = 0.0f;
  mma_mainloop_tn<Cfg>(acc, A, lda, m0, M, B, ldb, n0, N, r0, r1, smem, csa, csb, ca, cb);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  if (alpha_dev) alpha *= *alpha_dev;
  // column sums: the NTHREADS/ACH threads that staged the same columns are reduced through LDS, one atomic per column
  float* xs = reinterpret_cast<float*>(smem);
  if (csa) {
    const int grp = threadIdx.x / Cfg::ACH, ch = threadIdx.x % Cfg::ACH;
#pragma unroll
    for (int e = 0; e < Cfg::EPC; ++e) xs[grp * Cfg::BM + ch * Cfg::EPC + e] = ca[e];
    __syncthreads();
    if (threadIdx.x < Cfg::BM && m0 + threadIdx.x < M) {
      float t = 0.0f;
      for (int g = 0; g < Cfg::NTHREADS / Cfg::ACH; ++g) t += xs[g * Cfg::BM + threadIdx.x];
      atomicAdd(colsum_a + m0 + threadIdx.x, t * alpha);
    }
    __syncthreads();
  }
  if (csb) {
    const int grp = threadIdx.x / Cfg::BCH, ch = threadIdx.x % Cfg::BCH;
#pragma unroll
    for (int e = 0; e < Cfg::EPC; ++e) xs[grp * Cfg::BN + ch * Cfg::EPC + e] = cb[e];
    __syncthreads();
    if (threadIdx.x < Cfg::BN && n0 + threadIdx.x < N) {
      float t = 0.0f;
      for (int g = 0; g < Cfg::NTHREADS / Cfg::BCH; ++g) t += xs[g * Cfg::BN + threadIdx.x];
      atomicAdd(colsum_b + n0 + threadIdx.x, t * alpha);
    }
  }
  if (partial) {
    // two-stage split-K: this split's tile goes out with plain stores (two 128-byte row segments per wave instruction) into
    // its own slab [split][tile rows][tile cols]; tn_reduce_kernel sums the slabs into C.  The atomic form below moves the
    // same bytes at the chip's float-atomic rate (~1.3 TB/s): 40 us of a 107-us GEMM at 16 splits of a 1536 x 512 output.
    float* P = partial + ((long)split * tiles + tile) * (Cfg::BM * Cfg::BN);
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int ml = Cfg::row_of(wm, tm, reg, lane);
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) P[ml * Cfg::BN + Cfg::col_of(wn, tn, lane)] = acc[tm][tn][reg];
      }
    return;
  }
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

// ---------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM, large form (bf16, M and N multiples of 256): C[m, n] += sum_r A[r, m] B[r, n] with 256 x 256 output
// tiles, ONE 8-wave workgroup per CU, operands streamed by LDS-DMA through a 4-stage ring of 32-row tiles.
// Why: at 128 x 128 tiles a block moves 16 KB per 1 MFLOP (64 flop/B) and the measured L2->CU intake of a computing CU
// (~48 GB/s) caps the chip near 0.7 PFLOP/s whatever the occupancy; 256 x 256 tiles need half the bytes per flop.  The
// register-staged engine above cannot hold such a tile with enough blocks per CU to hide its loads; here the loads need no
// registers (LDS-DMA), three tiles (96 KB) are in flight per CU behind counted vmcnt waits, and the 128 accumulator registers
// per lane leave room for double-buffered fragments at two waves per SIMD.
// LDS image of an operand tile: [32 rows k][256 columns] bf16, rows unpadded (a DMA instruction is lane-linear: 64 lanes x
// 16 B = two whole rows), 16-byte chunk c of row k stored at position c ^ ((k & 3) << 2): the four consecutive k rows that
// one ds_read_b64_tr_b16 group touches then fall into four different bank quarters (conflict-free, checked by hand against
// the instruction's 2 x 32 lane groups).  The swizzle is applied on the SOURCE side: lane p of row k fetches chunk
// p ^ ((k & 3) << 2).  Rows beyond the split's range are fetched from a page of zeros (static counts of DMA instructions).
// Output: always the two-stage split-K form (slabs + tn_reduce_kernel).  No fused column sums: read from the LDS image they
// cost the blocks that owed them 60 % more time (the whole kernel waits for its slowest block); the decoder's bias
// gradients come out of the BPTT kernel instead (DecBwdPersist::dbhh), and a caller that wants column sums gets the
// 128 x 128 engine.
namespace tn256 {
constexpr int BM = 256, BN = 256, BKR = 32, NST = 4, NT = 512;
constexpr int ROWB = 512, OP_BYTES = BKR * ROWB, STAGE = 2 * OP_BYTES, SMEM = NST * STAGE;
static __device__ __attribute__((aligned(256))) unsigned char zero_page[512];   // never written: zero-initialised

__device__ __forceinline__ uint4 frag(const char* s, int kbase, int mbase, int lane) {
  // TnFrag<bf16_t>::load on the swizzled image
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int k = kbase + 8 * (g >> 1) + q;
  const int m = mbase + 16 * (g & 1) + 4 * p;
  typedef __attribute__((address_space(3))) s16x4* lptr;
  const char* a0 = s + k * ROWB + ((((m >> 3) ^ (q << 2))) << 4) + (m & 7) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * ROWB));
  uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}
}  // namespace tn256

__global__ __launch_bounds__(tn256::NT) void gemm_tn256_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B,
                                                               long ldb, int M, int N, int R, int rows_per_split, int nsplits,
                                                               float* __restrict__ partial) {
  using namespace tn256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
  // XCD-aware map (workgroups go round-robin over the 8 XCDs, each with its own L2): all tiles of a split on ONE XCD, so
  // that the split's rows of A and B cross the fabric once and the 2 / 6 re-reads by the other tiles are L2 hits
  const int spx = (nsplits + 7) / 8;   // splits per XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int split = xcd * spx + slot / tiles, tile = slot % tiles;
  if (split >= nsplits) return;
  const int bm = tile / tiles_n, bn = tile % tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int r0 = split * rows_per_split, r1 = min(R, r0 + rows_per_split);
  const int nk = (r1 - r0 + BKR - 1) / BKR;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // wave tile: 128 rows (m) x 64 columns (n)
  f32x16 acc[4][2];
  {
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = zacc;
  }

  // four DMA instructions per wave and tile: instruction i = wave + 8 j covers rows 2i', 2i'+1 of A (i < 16) or B
  auto issue = [&](int kt) __attribute__((always_inline)) {
    char* st = smem + (kt & (NST - 1)) * STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = wave + 8 * j;                 // wave-uniform; j < 2: A, j >= 2: B
      const bool isA = j < 2;
      const int pair = isA ? i : i - 16;
      const int row = 2 * pair + (lane >> 5), pos = lane & 31;
      const int c = pos ^ ((row & 3) << 2);
      const long grow = (long)r0 + (long)kt * BKR + row;
      const bf16_t* src = isA ? A + grow * lda + m0 + c * 8 : B + grow * ldb + n0 + c * 8;
      const void* s2 = grow < r1 ? (const void*)src : (const void*)(zero_page + pos * 16);
      pg_lds_dma16(s2, st + (isA ? 0 : OP_BYTES) + pair * 2 * ROWB);
    }
  };
  auto wait_tiles = [&](int younger) __attribute__((always_inline)) {   // all but the DMA of `younger` later tiles
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // hipcc must know that nothing it tracks is in flight when the loop starts (the DMA is invisible to it)
  __builtin_amdgcn_s_waitcnt(0);
  for (int p = 0; p < NST - 1; ++p)
    if (p < nk) issue(p);
  // Software pipeline over the tiles: the barrier that hands over tile kt+1 sits BETWEEN the two k-steps of tile kt, so
  // the fragments of (kt+1, step 0) are read while the MFMAs of (kt, step 1) run, and those of (kt, step 1) while the
  // hand-over waits: no k-step waits for its own LDS reads.  (Left to itself hipcc hoists all 24 fragment reads of a
  // tile in front of its 16 MFMAs and the wave idles for the LDS round trip once per tile.)
  // (two named register sets, selected at compile time: an array indexed by a lambda parameter ended up in scratch)
  uint4 fa0[4], fb0[2], fa1[4], fb1[2];
  auto lfrag = [&](int kt, auto set_c) __attribute__((always_inline)) {   // set s holds the fragments of k-step s
    constexpr int set = decltype(set_c)::value;
    const char* sa = smem + (kt & (NST - 1)) * STAGE;
    const char* sb = sa + OP_BYTES;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) (set ? fa1 : fa0)[tm] = frag(sa, set * 16, wm * 128 + tm * 32, lane);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) (set ? fb1 : fb0)[tn] = frag(sb, set * 16, wn * 64 + tn * 32, lane);
  };
  auto mma_set = [&](auto set_c) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) MmaOp<bf16_t>::run(acc[tm][tn], (set ? fa1 : fa0)[tm], (set ? fb1 : fb0)[tn]);
  };
  // the MFMAs of k-step `cur` with the fragment reads of (tile ktn, k-step 1 - cur) issued between them, two MFMAs then
  // three reads at a time: measured alone the kernel's MFMAs take 81 us and its LDS reads 48 - back to back they do not
  // overlap inside a wave, and the second wave of the SIMD is in the same phase (the tile barrier keeps them in step)
  auto mma_and_read = [&](auto cur_c, int ktn) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_c)::value, nxt = 1 - cur;
    const char* sa = smem + (ktn & (NST - 1)) * STAGE;
    const char* sb = sa + OP_BYTES;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      MmaOp<bf16_t>::run(acc[tm][0], (cur ? fa1 : fa0)[tm], (cur ? fb1 : fb0)[0]);
      MmaOp<bf16_t>::run(acc[tm][1], (cur ? fa1 : fa0)[tm], (cur ? fb1 : fb0)[1]);
      __builtin_amdgcn_sched_barrier(0);
      (nxt ? fa1 : fa0)[tm] = frag(sa, nxt * 16, wm * 128 + tm * 32, lane);
      if (tm < 2) (nxt ? fb1 : fb0)[tm] = frag(sb, nxt * 16, wn * 64 + tm * 32, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  static_assert(BKR == 32, "two k-steps per tile");
  wait_tiles(nk - 1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  lfrag(0, IntC<0>{});
  // Order inside a tile: the MFMAs of step 0 first (their fragments were read half a tile ago), THEN the 12 reads of step 1
  // - read first, 24 LDS operations would be outstanding in front of the MFMAs, more than the 4-bit lgkmcnt can count, and
  // hipcc falls back to lgkmcnt(0).  (Spreading the reads between the MFMAs with sched_group_barrier spilled.)
#pragma unroll 1
  for (int kt = 0; kt + 1 < nk; ++kt) {
    __builtin_amdgcn_sched_barrier(0);
    mma_and_read(IntC<0>{}, kt);          // k-step 0 of tile kt; reads k-step 1 of tile kt
    // ---- hand-over of tile kt+1: this wave's DMA of it has landed, the barrier extends that to every wave's and proves
    // that nobody reads tile kt-1 any more, whose slot the next request overwrites
    wait_tiles(nk - 2 - kt < 1 ? nk - 2 - kt : 1);   // tiles requested after kt+1 so far: kt+2 at most
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    mma_and_read(IntC<1>{}, kt + 1);      // k-step 1 of tile kt; reads k-step 0 of tile kt+1
    // the next request goes out behind the MFMAs: its address arithmetic runs under them
    if (kt + NST - 1 < nk) issue(kt + NST - 1);
  }
  __builtin_amdgcn_sched_barrier(0);
  mma_set(IntC<0>{});
  lfrag(nk - 1, IntC<1>{});
  mma_set(IntC<1>{});
  // ---- this split's tile as a slab (tn_reduce_kernel<256, 256> sums the slabs into C)
  float* P = partial + ((long)split * tiles + tile) * (BM * BN);
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int ml = wm * 128 + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) P[ml * BN + wn * 64 + tn * 32 + (lane & 31)] = acc[tm][tn][reg];
    }
}

// C[m, n] += alpha * sum over splits of the slabs written by gemm_tn_kernel; one thread per 4 columns of one tile row
template <int BM, int BN>
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ partial, int nsplits, int tiles, int tiles_n,
                                                         int M, int N, float* __restrict__ C, long ldc, float alpha,
                                                         const float* __restrict__ alpha_dev) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int Q = BN / 4;
  const long per_tile = (long)BM * Q;
  if (id >= per_tile * tiles) return;
  const int tile = (int)(id / per_tile), rem = (int)(id % per_tile);
  const int ml = rem / Q, q = rem % Q;
  const float4* src = reinterpret_cast<const float4*>(partial + (long)tile * (BM * BN) + (long)ml * BN) + q;
  const long sstride = (long)tiles * (BM * BN) / 4;
  float4 s = src[0];
  for (int k = 1; k < nsplits; ++k) {
    const float4 v = src[k * sstride];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (alpha_dev) alpha *= *alpha_dev;
  const int m = (tile / tiles_n) * BM + ml, n = (tile % tiles_n) * BN + q * 4;
  if (m >= M) return;
  float* c = C + (long)m * ldc + n;
  if (n + 3 < N && ((reinterpret_cast<uintptr_t>(c) & 15) == 0)) {
    float4 o = *reinterpret_cast<float4*>(c);
    o.x += alpha * s.x; o.y += alpha * s.y; o.z += alpha * s.z; o.w += alpha * s.w;
    *reinterpret_cast<float4*>(c) = o;
  } else {
    const float v[4] = {s.x, s.y, s.z, s.w};
    for (int e = 0; e < 4; ++e) if (n + e < N) c[e] += alpha * v[e];
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
int launch_gemm_nt_splitk(const T* A, long lda, const T* B, long ldb, int M, int N, int K, float* C, long ldc, float alpha,
                          hipStream_t stream) {
  constexpr int EPC = 16 / sizeof(T);
  if (M <= 0 || N <= 0 || K <= 0) return PG_OK;
  PG_REQUIRE(K % EPC == 0 && lda % EPC == 0 && ldb % EPC == 0, "gemm_nt_splitk: K/lda/ldb (%d,%ld,%ld) must be multiples of %d",
             K, lda, ldb, EPC);
  PG_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && C, "gemm_nt_splitk: operands must be 16-byte aligned");
  // Two tilings.  These GEMMs sit on the chain between the decoder BPTT and the encoder BPTT while the 256 x 256 weight-gradient
  // kernels hold every CU (one 128-KB workgroup each, 2 x 192 of them queued): a 128 x 128 block wants 73 KB of LDS, finds none
  // beside them and waits for a CU to come free (15 us alone, up to 80 us in the step).  64 x 64 tiles with 32-element K
  // steps (PG_SPLITK_TILE=64) take 20 KB and 48 registers, fit BESIDE a resident 256 x 256 workgroup and run in 13 + 11 us
  // there - but the wait then falls to the next LDS-heavy kernel of the chain and the step is unchanged (measured), so the
  // default stays the larger tile.
  static const int tile_env = getenv("PG_SPLITK_TILE") ? atoi(getenv("PG_SPLITK_TILE")) : 0;
  const bool small = tile_env == 64;
  if (small) {
    using Cfg = TileCfg<T, 64, 64, 4, 2, 2>;
    const int tiles = cdiv(M, Cfg::BM) * cdiv(N, Cfg::BN), nkt = cdiv(K, Cfg::BK);
    int splits = max(1, min(nkt / 2, 256 / max(1, tiles)));
    const int kt_per_split = cdiv(nkt, splits);
    splits = cdiv(nkt, kt_per_split);
    PG_LAUNCH_BIG_SMEM(gemm_nt_splitk_kernel<Cfg>, dim3(tiles * splits), dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, A, lda, B,
                       ldb, M, N, K, kt_per_split, C, ldc, alpha);
    PG_CHECK_LAUNCH();
    return PG_OK;
  }
  using Cfg = TileCfg<T, 128, 128, 8, 2, 2>;
  const int tiles = cdiv(M, Cfg::BM) * cdiv(N, Cfg::BN), nkt = cdiv(K, Cfg::BK);
  // about one block per CU, at least 2 K tiles per block
  int splits = max(1, min(nkt / 2, 256 / max(1, tiles)));
  const int kt_per_split = cdiv(nkt, splits);
  splits = cdiv(nkt, kt_per_split);
  PG_LAUNCH_BIG_SMEM(gemm_nt_splitk_kernel<Cfg>, dim3(tiles * splits), dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, A, lda, B, ldb,
                     M, N, K, kt_per_split, C, ldc, alpha);
  PG_CHECK_LAUNCH();
  return PG_OK;
}

template <class Cfg, typename T>
static int launch_gemm_tn_cfg(const T* A, long lda, const T* B, long ldb, int M, int N, int R, float* C, long ldc, float alpha,
                              const float* alpha_dev, int target_blocks, float* colsum_a, float* colsum_b, hipStream_t stream,
                              float* partial, size_t partial_bytes) {
  static_assert(Cfg::SMEM_BYTES >= (Cfg::NTHREADS / Cfg::ACH) * Cfg::BM * 4 && Cfg::SMEM_BYTES >= (Cfg::NTHREADS / Cfg::BCH) * Cfg::BN * 4,
                "column-sum reduction must fit the staging LDS");
  const int tiles = cdiv(M, Cfg::BM) * cdiv(N, Cfg::BN);
  int splits = max(1, min(cdiv(R, Cfg::BKR * 4), max(1, target_blocks / tiles)));
  if (splits > 8) splits -= splits % 8;   // whole rounds over the 8 XCDs (see the kernel's block map)
  int rows_per_split = cdiv(cdiv(R, splits), Cfg::BKR) * Cfg::BKR;
  splits = cdiv(R, rows_per_split);
  dim3 grid(8 * cdiv(splits, 8) * tiles);
  // two-stage reduction when the caller lent a slab buffer that holds every split's tiles and the output is large enough
  // for the atomics to matter (the column sums stay atomic: a few KB)
  const size_t need = (size_t)splits * tiles * Cfg::BM * Cfg::BN * sizeof(float);
  // (with a device-side scale the fused column sums are accumulated inside the GEMM kernel, which on the two-stage path
  // no longer sees alpha_dev: that combination takes the atomic path instead of producing unscaled bias gradients)
  const bool two_stage = partial && need <= partial_bytes && splits >= 4 && tiles >= 16 && !(alpha_dev && (colsum_a || colsum_b));
  PG_LAUNCH_BIG_SMEM(gemm_tn_kernel<Cfg>, grid, dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, A, lda, B, ldb, M, N, R,
                     rows_per_split, splits, C, ldc, alpha, two_stage ? nullptr : alpha_dev, colsum_a, colsum_b,
                     two_stage ? partial : (float*)nullptr);
  PG_CHECK_LAUNCH();
  if (two_stage) {
    const long threads = (long)tiles * Cfg::BM * (Cfg::BN / 4);
    hipLaunchKernelGGL((tn_reduce_kernel<Cfg::BM, Cfg::BN>), dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, stream, partial, splits,
                       tiles, cdiv(N, Cfg::BN), M, N, C, ldc, alpha, alpha_dev);
    PG_CHECK_LAUNCH();
  }
  return PG_OK;
}

// workgroups the next 256 x 256 launches aim for (0: the default); set around calls by pg_vae.hip, one calling thread
int g_tn256_blocks_hint = 0;

template <typename T>
int launch_gemm_tn(const T* A, long lda, const T* B, long ldb, int M, int N, int R, float* C, long ldc, float alpha,
                   const float* alpha_dev, hipStream_t stream, float* colsum_a, float* colsum_b, float* partial,
                   size_t partial_bytes) {
  constexpr int EPC = 16 / sizeof(T);
  if (M <= 0 || N <= 0 || R <= 0) return PG_OK;
  PG_REQUIRE(lda % EPC == 0 && ldb % EPC == 0, "gemm_tn: lda/ldb must be multiples of %d", EPC);
  PG_REQUIRE(lda >= (M + EPC - 1) / EPC * EPC && ldb >= (N + EPC - 1) / EPC * EPC, "gemm_tn: rows must be padded to whole chunks");
  PG_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_tn: operands must be 16-byte aligned");
  // Measured on the decoder's weight-gradient shape (1536 x 512 x 50 k rows, bf16): 64-row tiles, one tile in flight, 384
  // blocks 133 us; 32-row tiles (40 KB of LDS: three blocks per CU), two tiles in flight, 768 blocks 108 us; deeper rings
  // and 256-wide tiles are slower (registers / LDS cut the blocks per CU).  Few output tiles: fewer split-K blocks, whose
  // fp32 atomics otherwise dominate (768 x 256: 48 us at 384 blocks, 59 us at 768).
  if constexpr (std::is_same<T, bf16_t>::value) {
    // large outputs: 256 x 256 tiles, one workgroup per CU, LDS-DMA ring (gemm_tn256_kernel); 192 blocks leave 64 CUs to
    // the encoder BPTT that runs beside these GEMMs in the training step
    static const int tn256 = getenv("PG_TN256") ? atoi(getenv("PG_TN256")) : 1;
    if (tn256 && partial && !alpha_dev && !colsum_a && !colsum_b && M % 256 == 0 && N % 256 == 0 && (M / 256) * (N / 256) >= 8) {
      const int tiles = (M / 256) * (N / 256);
      static const int blocks_env = getenv("PG_TN256_BLOCKS") ? atoi(getenv("PG_TN256_BLOCKS")) : 192;
      const int want = g_tn256_blocks_hint > 0 ? g_tn256_blocks_hint : blocks_env;
      int splits = max(1, want / tiles);
      int rows_per_split = cdiv(cdiv(R, splits), 32) * 32;
      splits = cdiv(R, rows_per_split);
      if ((size_t)splits * tiles * 256 * 256 * sizeof(float) <= partial_bytes) {
        PG_LAUNCH_BIG_SMEM(gemm_tn256_kernel, dim3(8 * cdiv(splits, 8) * tiles), dim3(tn256::NT), tn256::SMEM, stream, A, lda, B, ldb, M, N, R,
                           rows_per_split, splits, partial);
        PG_CHECK_LAUNCH();
        const long threads = (long)tiles * 256 * (256 / 4);
        hipLaunchKernelGGL((tn_reduce_kernel<256, 256>), dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, stream, partial, splits,
                           tiles, N / 256, M, N, C, ldc, alpha, (const float*)nullptr);
        PG_CHECK_LAUNCH();
        return PG_OK;
      }
    }
  }
  static const int tn_blocks_env = getenv("PG_TN_BLOCKS") ? atoi(getenv("PG_TN_BLOCKS")) : 0;   // tuning aids
  static const int tn_bkr = getenv("PG_TN_BKR") ? atoi(getenv("PG_TN_BKR")) : 32;
  static const int tn_depth = getenv("PG_TN_DEPTH") ? atoi(getenv("PG_TN_DEPTH")) : 2;
  // (with slabs instead of atomics the large shape is 20 us per step better at 384 blocks: half the slab bytes)
  const int tn_blocks = tn_blocks_env ? tn_blocks_env : (cdiv(M, 128) * cdiv(N, 128) >= 32 && !partial ? 768 : 384);
#define PG_TN_CASE(BKR, DEP)                                                                                                    \
  if (tn_bkr == BKR && tn_depth == DEP)                                                                                         \
    return launch_gemm_tn_cfg<TileTN<T, 128, 128, BKR, 2, 2, DEP>>(A, lda, B, ldb, M, N, R, C, ldc, alpha, alpha_dev, tn_blocks, \
                                                                  colsum_a, colsum_b, stream, partial, partial_bytes);
  PG_TN_CASE(64, 1) PG_TN_CASE(32, 1) PG_TN_CASE(32, 2)
#undef PG_TN_CASE
  return launch_gemm_tn_cfg<TileTN<T, 128, 128, 32, 2, 2, 2>>(A, lda, B, ldb, M, N, R, C, ldc, alpha, alpha_dev, tn_blocks, colsum_a,
                                                              colsum_b, stream, partial, partial_bytes);
}

template int launch_gemm_nt<float>(const float*, long, const float*, long, int, int, int, const GemmEpi<float>&, hipStream_t);
template int launch_gemm_nt<bf16_t>(const bf16_t*, long, const bf16_t*, long, int, int, int, const GemmEpi<bf16_t>&, hipStream_t);
template int launch_gemm_tn<float>(const float*, long, const float*, long, int, int, int, float*, long, float, const float*, hipStream_t, float*, float*, float*, size_t);
template int launch_gemm_tn<bf16_t>(const bf16_t*, long, const bf16_t*, long, int, int, int, float*, long, float, const float*, hipStream_t, float*, float*, float*, size_t);
template int launch_gemm_nt_splitk<float>(const float*, long, const float*, long, int, int, int, float*, long, float, hipStream_t);
template int launch_gemm_nt_splitk<bf16_t>(const bf16_t*, long, const bf16_t*, long, int, int, int, float*, long, float, hipStream_t);
