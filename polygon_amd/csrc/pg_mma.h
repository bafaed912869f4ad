// MFMA tile engines for gfx950 (64-wide waves, 32x32 MFMA tiles, LDS-staged operands).
//
//  * TileCfg / mma_mainloop_nt : C[m,n] += sum_k A[m,k] * B[n,k]   (both operands K-contiguous: activations x
//    torch-layout weights).  Operands are staged global -> registers -> LDS in 16-byte chunks, double buffered,
//    one barrier per K tile.  LDS rows are padded by 16 B so that the ds_read_b128 fragment reads are conflict free.
//    bf16: v_mfma_f32_32x32x16_bf16 (fragment = 8 consecutive k in one 16-B chunk);
//    f32 : v_mfma_f32_32x32x2_f32 x4 per 16-B chunk (lane half h supplies k = 4h..4h+3 of each 8-k step; A and B use
//          the same assignment, so the contraction is complete and exact f32).
//  * TileTN / mma_mainloop_tn : C[m,n] += sum_r A[r,m] * B[r,n]    (contraction over the ROW index of two row-major
//    activations: weight gradients).  LDS keeps the tiles as they lie in memory ([k][m]); bf16 fragments are read
//    with the gfx950 hardware-transposing ds_read_b64_tr_b16, f32 fragments with plain ds_read_b32.
#pragma once
#include "pg_common.h"

template <int I> struct IntC { static constexpr int value = I; };
template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<N, I + 1>(f);
  }
}

// Branch-free predicated 16-byte global load: a raw buffer descriptor over [base, base + 2 GiB) and a per-lane byte
// offset; lanes that must not load pass PG_OOB, which the hardware range check turns into zeros (no branch, so hipcc
// keeps counted vmcnt waits and the register prefetch ring really stays in flight).
#define PG_OOB 0xFFFFFFF0u
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pg_make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ uint4 pg_buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// LDS-DMA by inline asm: 16 bytes per lane from a per-lane global address, lane-linear at the wave-uniform LDS address
// `dst`.  (The builtin form makes hipcc track the transfer and guard later LDS reads with waits of its own; issued from
// asm the transfers are invisible to it and every wait is the kernel's own counted one.)
__device__ __forceinline__ void pg_lds_dma16(const void* src, void* dst) {
  const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m) : "memory", "m0");
}

typedef __attribute__((ext_vector_type(4))) float f32x4;

// Four K-steps x three gates as ONE asm statement (12 MFMAs; accumulators %0-%2, weights %3-%14 gate-major, activation
// fragments %15-%18): hipcc then waits once for the group's LDS fragments instead of once per single-MFMA asm, and the
// per-MFMA nop goes (one wave per SIMD issues one instruction at a time: the 96 nops of a tile were an eighth of its
// instructions).  Accumulation on one destination three MFMAs apart is interlocked by the hardware.  The `s_nop 0` between
// the MFMAs is NOT a hazard nop: a lone wave that issues 16x16x32 MFMAs back to back gets one per 45 cycles (18.8 ns), with
// any one instruction in between one per 33 cycles (13.8 ns; tools/mfma_rate.hip) - the matrix pipe's own rate.  FIRST: the tile's
// first group, behind the VALU moves that zero the accumulators.  CA/CB/CC: register class of each gate's weights.
#define PG_MMA12_BODY                                              \
  "v_mfma_f32_16x16x32_bf16 %0, %3, %15, %0\n\ts_nop 0\n\t"      \
  "v_mfma_f32_16x16x32_bf16 %1, %7, %15, %1\n\ts_nop 0\n\t"      \
  "v_mfma_f32_16x16x32_bf16 %2, %11, %15, %2\n\ts_nop 0\n\t"     \
  "v_mfma_f32_16x16x32_bf16 %0, %4, %16, %0\n\ts_nop 0\n\t"      \
  "v_mfma_f32_16x16x32_bf16 %1, %8, %16, %1\n\ts_nop 0\n\t"      \
  "v_mfma_f32_16x16x32_bf16 %2, %12, %16, %2\n\ts_nop 0\n\t"     \
  "v_mfma_f32_16x16x32_bf16 %0, %5, %17, %0\n\ts_nop 0\n\t"      \
  "v_mfma_f32_16x16x32_bf16 %1, %9, %17, %1\n\ts_nop 0\n\t"      \
  "v_mfma_f32_16x16x32_bf16 %2, %13, %17, %2\n\ts_nop 0\n\t"     \
  "v_mfma_f32_16x16x32_bf16 %0, %6, %18, %0\n\ts_nop 0\n\t"      \
  "v_mfma_f32_16x16x32_bf16 %1, %10, %18, %1\n\ts_nop 0\n\t"     \
  "v_mfma_f32_16x16x32_bf16 %2, %14, %18, %2\n\ts_nop 0"
#define PG_MMA12_OPS(CA, CB, CC)                                                                                         \
  : "+v"(c0), "+v"(c1), "+v"(c2)                                                                                         \
  : CA(wa[0]), CA(wa[1]), CA(wa[2]), CA(wa[3]), CB(wb[0]), CB(wb[1]), CB(wb[2]), CB(wb[3]), CC(wc[0]), CC(wc[1]), CC(wc[2]), \
    CC(wc[3]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3])
// NOP: the statement starts with `s_nop 1` - for a tile's first group (its accumulators were just zeroed by VALU moves, and
// a VALU write needs two wait states before an MFMA reads the register) and for EVERY group of the instrumented
// instantiations, where hipcc sets up operands with VALU moves right in front of a statement (tools/check_mfma_hazard.py,
// run by tests/test_isa_hazards.py over every instantiation, found one there).  In the production instantiations a nop per
// group measured +10 us per kernel, so they carry it only where the listing check needs it.
// all three gates' weights in AGPRs (the recurrent half)
template <bool NOP>
__device__ __forceinline__ void mma12_aaa(f32x4& c0, f32x4& c1, f32x4& c2, const bf16x8* wa, const bf16x8* wb, const bf16x8* wc,
                                          const bf16x8* x) {
  if constexpr (NOP) asm("s_nop 1\n\t" PG_MMA12_BODY PG_MMA12_OPS("a", "a", "a"));
  else asm(PG_MMA12_BODY PG_MMA12_OPS("a", "a", "a"));
}
// input half: r weights in AGPRs, z weights in VGPRs, n weights read from LDS
template <bool NOP>
__device__ __forceinline__ void mma12_avv(f32x4& c0, f32x4& c1, f32x4& c2, const bf16x8* wa, const bf16x8* wb, const bf16x8* wc,
                                          const bf16x8* x) {
  if constexpr (NOP) asm("s_nop 1\n\t" PG_MMA12_BODY PG_MMA12_OPS("a", "v", "v"));
  else asm(PG_MMA12_BODY PG_MMA12_OPS("a", "v", "v"));
}

template <typename T> struct MmaOp;
template <> struct MmaOp<bf16_t> {
  static __device__ __forceinline__ void run(f32x16& c, const uint4& a, const uint4& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct MmaOp<float> {
  static __device__ __forceinline__ void run(f32x16& c, const uint4& a, const uint4& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// Row functors: element offset (from .base) of tile row `r` (K-contiguous), or -1 for an out-of-range row.
template <typename T> struct RowSrc {
  const T* base;
  long ld;
  int row0, nrows;
  __device__ __forceinline__ long operator()(int r) const {   // element offset of tile row r, -1 if out of range
    int g = row0 + r;
    return g < nrows ? (long)g * ld : -1L;
  }
};
// Weight rows of a GRU: tile row n = gate*BH + jj maps to weight row gate*H + j0 + jj.
template <typename T> struct GateRowSrc {
  const T* base;
  long ld;
  int H, j0, BH;
  __device__ __forceinline__ long operator()(int n) const {
    int g = n / BH, j = j0 + (n - g * BH);
    return j < H ? (long)(g * H + j) * ld : -1L;
  }
};

template <typename T_, int BM_, int BN_, int BKC_, int WM_, int WN_, int NGATE_ = 1, int DEPTH_ = 1> struct TileCfg {
  using T = T_;
  static constexpr int BM = BM_, BN = BN_, BKC = BKC_, WM = WM_, WN = WN_, NGATE = NGATE_;
  static constexpr int DEPTH = DEPTH_;         // K tiles kept in flight in registers (global-load prefetch depth)
  static constexpr int NTHREADS = 64 * WM * WN;
  static constexpr int EPC = 16 / sizeof(T);   // elements per 16-byte chunk
  static constexpr int BK = BKC * EPC;         // K elements per tile
  static constexpr int ROWB = BKC * 16 + 16;   // padded LDS row bytes
  static constexpr int WTM = BM / WM;          // rows per wave
  static constexpr int TM = WTM / 32;
  static constexpr int BNG = BN / NGATE;       // block columns per gate group
  static constexpr int WTN = BNG / WN;         // columns per wave per gate group
  static constexpr int TU = WTN / 32;
  static constexpr int TN = TU * NGATE;
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int SMEM_BYTES = 2 * STAGE_BYTES;
  static constexpr int A_IT = (BM * BKC + NTHREADS - 1) / NTHREADS;
  static constexpr int B_IT = (BN * BKC + NTHREADS - 1) / NTHREADS;
  static_assert(BKC % 2 == 0, "BKC must be even");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32");
  // block-local column (within its gate group) of accumulator tile tn, lane column c
  static __device__ __forceinline__ int col_in_group(int wn, int tn, int c) { return wn * WTN + (tn % TU) * 32 + c; }
  static __device__ __forceinline__ int gate_of(int tn) { return tn / TU; }
  static __device__ __forceinline__ int row_of(int wm, int tm, int reg, int lane) {
    return wm * WTM + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
  }
};

// acc[tm][slot(tn)] += A-tile x B-tile over K.  SLOT_LAST: accumulator slot used by the LAST gate group (lets a
// second pass route the n-gate to a separate accumulator).  Leaves the block synchronised (LDS reusable).
template <class Cfg, int NACC, int SLOT_LAST, class ASrc, class BSrc>
__device__ __forceinline__ void mma_mainloop_nt(f32x16 (&acc)[Cfg::TM][NACC], const ASrc& asrc, const BSrc& bsrc,
                                                int K, char* smem, int tid, int kt_begin_, int nk) {
  const int kt_begin = __builtin_amdgcn_readfirstlane(kt_begin_);
  // tid: thread index inside the K-group (Cfg::NTHREADS threads share `smem`); the group contracts K tiles
  // kt_begin .. kt_begin+nk-1.  Every K-group of a block must pass the same nk (the barriers are block-wide).
  using T = typename Cfg::T;
  constexpr int NT = Cfg::NTHREADS, BKC = Cfg::BKC, EPC = Cfg::EPC, ROWB = Cfg::ROWB;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const __amdgpu_buffer_rsrc_t arsrc = pg_make_rsrc(asrc.base);
  const __amdgpu_buffer_rsrc_t brsrc = pg_make_rsrc(bsrc.base);
  unsigned pa[Cfg::A_IT], pb[Cfg::B_IT];   // per-lane byte offset of its chunk at k = 0 (PG_OOB: never loads)
  int oa[Cfg::A_IT], ob[Cfg::B_IT];
#pragma unroll
  for (int i = 0; i < Cfg::A_IT; ++i) {
    int id = tid + i * NT;
    int row = id / BKC, c = id % BKC;
    bool ok = id < Cfg::BM * BKC;
    long o = ok ? asrc(row) : -1L;
    pa[i] = o >= 0 ? (unsigned)((o + c * EPC) * (long)sizeof(T)) : PG_OOB;
    oa[i] = ok ? row * ROWB + c * 16 : -1;
  }
#pragma unroll
  for (int i = 0; i < Cfg::B_IT; ++i) {
    int id = tid + i * NT;
    int row = id / BKC, c = id % BKC;
    bool ok = id < Cfg::BN * BKC;
    long o = ok ? bsrc(row) : -1L;
    pb[i] = o >= 0 ? (unsigned)((o + c * EPC) * (long)sizeof(T)) : PG_OOB;
    ob[i] = ok ? row * ROWB + c * 16 : -1;
  }
  constexpr int D = Cfg::DEPTH;
  uint4 ra[D][Cfg::A_IT], rb[D][Cfg::B_IT];   // register ring: tile j lives in slot j % D (static indices only)
  const int kend = min(K, (kt_begin + nk) * Cfg::BK);   // this K-group's end: nothing is fetched past it
  // chunks past kend (and whole tiles past the last one) get PG_OOB: zeros, no memory traffic, no branch
  auto gload = [&](int kt, auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    const int k0 = (kt_begin + kt) * Cfg::BK;
    const unsigned soff = (unsigned)k0 * (unsigned)sizeof(T);
#pragma unroll
    for (int i = 0; i < Cfg::A_IT; ++i) {
      int c = (tid + i * NT) % BKC;
      ra[slot][i] = pg_buf_load16(arsrc, (k0 + c * EPC < kend) ? pa[i] : PG_OOB, soff);
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_IT; ++i) {
      int c = (tid + i * NT) % BKC;
      rb[slot][i] = pg_buf_load16(brsrc, (k0 + c * EPC < kend) ? pb[i] : PG_OOB, soff);
    }
  };
  auto sstore = [&](int buf, auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    char* sa = smem + buf * Cfg::STAGE_BYTES;
    char* sb = sa + Cfg::A_BYTES;
#pragma unroll
    for (int i = 0; i < Cfg::A_IT; ++i)
      if (oa[i] >= 0) *reinterpret_cast<uint4*>(sa + oa[i]) = ra[slot][i];
#pragma unroll
    for (int i = 0; i < Cfg::B_IT; ++i)
      if (ob[i] >= 0) *reinterpret_cast<uint4*>(sb + ob[i]) = rb[slot][i];
  };
  const int lr = lane & 31, lh = lane >> 5;
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* sa = smem + buf * Cfg::STAGE_BYTES;
    const char* sb = sa + Cfg::A_BYTES;
#pragma unroll
    for (int s = 0; s < BKC / 2; ++s) {
      uint4 fa[Cfg::TM], fb[Cfg::TN];
#pragma unroll
      for (int tm = 0; tm < Cfg::TM; ++tm)
        fa[tm] = *reinterpret_cast<const uint4*>(sa + (wm * Cfg::WTM + tm * 32 + lr) * ROWB + (s * 2 + lh) * 16);
#pragma unroll
      for (int tn = 0; tn < Cfg::TN; ++tn) {
        int brow = Cfg::gate_of(tn) * Cfg::BNG + Cfg::col_in_group(wn, tn, lr);
        fb[tn] = *reinterpret_cast<const uint4*>(sb + brow * ROWB + (s * 2 + lh) * 16);
      }
#pragma unroll
      for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) {
          constexpr int TU = Cfg::TU;
          const int slot = (tn / TU == Cfg::NGATE - 1) ? SLOT_LAST * TU + tn % TU : tn;
          MmaOp<T>::run(acc[tm][slot], fa[tm], fb[tn]);
        }
    }
  };
  // prologue: tiles 0..D-1 in flight, tile 0 staged
  static_for<D>([&](auto d) __attribute__((always_inline)) { gload(decltype(d)::value, d); });
  sstore(0, IntC<0>{});
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += D) {
    static_for<D>([&](auto d) __attribute__((always_inline)) {
      constexpr int dd = decltype(d)::value;
      const int kt = kt0 + dd;
      if (kt < nk) {                               // block-uniform
        gload(kt + D, d);                          // slot dd was staged one iteration ago: refill it
        compute(kt & 1);
        sstore((kt + 1) & 1, IntC<(dd + 1) % D>{});  // tile kt+1 (oldest loads in flight) -> idle LDS buffer
        __syncthreads();
      }
    });
  }
}

// Two operand pairs contracted in ONE pipelined K loop (no second pipeline fill): virtual K tiles 0..nk1-1 come from
// (a1, b1) with the last gate routed to accumulator slot SLOT1, tiles nk1.. from (a2, b2) routed to SLOT2.  Used by the
// fused GRU cells: h.W_hh^T then x.W_ih^T (forward), dGH.W_hh then dGI_up.W_ih_up (backward).  K2 = 0 disables pair 2.
// The group contracts virtual tiles kt_begin .. kt_begin+nk-1 (same nk for every K-group of the block).
template <class Cfg, int NACC, int SLOT1, int SLOT2, class ASrc1, class BSrc1, class ASrc2, class BSrc2>
__device__ __forceinline__ void mma_mainloop_nt2(f32x16 (&acc)[Cfg::TM][NACC], const ASrc1& a1, const BSrc1& b1, int K1,
                                                 const ASrc2& a2, const BSrc2& b2, int K2, char* smem, int tid,
                                                 int kt_begin_, int nk) {
  using T = typename Cfg::T;
  constexpr int NT = Cfg::NTHREADS, BKC = Cfg::BKC, EPC = Cfg::EPC, ROWB = Cfg::ROWB;
  const int kt_begin = __builtin_amdgcn_readfirstlane(kt_begin_);   // keeps descriptor selects / soffsets scalar
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int nk1 = (K1 + Cfg::BK - 1) / Cfg::BK;
  const __amdgpu_buffer_rsrc_t ar1 = pg_make_rsrc(a1.base), br1 = pg_make_rsrc(b1.base);
  const __amdgpu_buffer_rsrc_t ar2 = pg_make_rsrc(K2 > 0 ? a2.base : a1.base), br2 = pg_make_rsrc(K2 > 0 ? b2.base : b1.base);
  unsigned pa1[Cfg::A_IT], pb1[Cfg::B_IT], pa2[Cfg::A_IT], pb2[Cfg::B_IT];
  int oa[Cfg::A_IT], ob[Cfg::B_IT];
#pragma unroll
  for (int i = 0; i < Cfg::A_IT; ++i) {
    int id = tid + i * NT;
    int row = id / BKC, c = id % BKC;
    bool ok = id < Cfg::BM * BKC;
    long o1 = ok ? a1(row) : -1L, o2 = (ok && K2 > 0) ? a2(row) : -1L;
    pa1[i] = o1 >= 0 ? (unsigned)((o1 + c * EPC) * (long)sizeof(T)) : PG_OOB;
    pa2[i] = o2 >= 0 ? (unsigned)((o2 + c * EPC) * (long)sizeof(T)) : PG_OOB;
    oa[i] = ok ? row * ROWB + c * 16 : -1;
  }
#pragma unroll
  for (int i = 0; i < Cfg::B_IT; ++i) {
    int id = tid + i * NT;
    int row = id / BKC, c = id % BKC;
    bool ok = id < Cfg::BN * BKC;
    long o1 = ok ? b1(row) : -1L, o2 = (ok && K2 > 0) ? b2(row) : -1L;
    pb1[i] = o1 >= 0 ? (unsigned)((o1 + c * EPC) * (long)sizeof(T)) : PG_OOB;
    pb2[i] = o2 >= 0 ? (unsigned)((o2 + c * EPC) * (long)sizeof(T)) : PG_OOB;
    ob[i] = ok ? row * ROWB + c * 16 : -1;
  }
  constexpr int D = Cfg::DEPTH;
  uint4 ra[D][Cfg::A_IT], rb[D][Cfg::B_IT];
  const int vend = kt_begin + nk;
  auto gload = [&](int kt, auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    const int v = kt_begin + kt;                 // virtual tile (block-uniform)
    const bool seg2 = v >= nk1;
    const int k0 = (seg2 ? v - nk1 : v) * Cfg::BK;
    const int kend = v < vend ? (seg2 ? K2 : K1) : 0;
    const unsigned soff = (unsigned)k0 * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t ar = seg2 ? ar2 : ar1, br = seg2 ? br2 : br1;
#pragma unroll
    for (int i = 0; i < Cfg::A_IT; ++i) {
      int c = (tid + i * NT) % BKC;
      ra[slot][i] = pg_buf_load16(ar, (k0 + c * EPC < kend) ? (seg2 ? pa2[i] : pa1[i]) : PG_OOB, soff);
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_IT; ++i) {
      int c = (tid + i * NT) % BKC;
      rb[slot][i] = pg_buf_load16(br, (k0 + c * EPC < kend) ? (seg2 ? pb2[i] : pb1[i]) : PG_OOB, soff);
    }
  };
  auto sstore = [&](int buf, auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    char* sa = smem + buf * Cfg::STAGE_BYTES;
    char* sb = sa + Cfg::A_BYTES;
#pragma unroll
    for (int i = 0; i < Cfg::A_IT; ++i)
      if (oa[i] >= 0) *reinterpret_cast<uint4*>(sa + oa[i]) = ra[slot][i];
#pragma unroll
    for (int i = 0; i < Cfg::B_IT; ++i)
      if (ob[i] >= 0) *reinterpret_cast<uint4*>(sb + ob[i]) = rb[slot][i];
  };
  const int lr = lane & 31, lh = lane >> 5;
  auto compute = [&](int buf, auto slot_last_c) __attribute__((always_inline)) {
    constexpr int SLOT_LAST = decltype(slot_last_c)::value;
    const char* sa = smem + buf * Cfg::STAGE_BYTES;
    const char* sb = sa + Cfg::A_BYTES;
#pragma unroll
    for (int s = 0; s < BKC / 2; ++s) {
      uint4 fa[Cfg::TM], fb[Cfg::TN];
#pragma unroll
      for (int tm = 0; tm < Cfg::TM; ++tm)
        fa[tm] = *reinterpret_cast<const uint4*>(sa + (wm * Cfg::WTM + tm * 32 + lr) * ROWB + (s * 2 + lh) * 16);
#pragma unroll
      for (int tn = 0; tn < Cfg::TN; ++tn) {
        int brow = Cfg::gate_of(tn) * Cfg::BNG + Cfg::col_in_group(wn, tn, lr);
        fb[tn] = *reinterpret_cast<const uint4*>(sb + brow * ROWB + (s * 2 + lh) * 16);
      }
#pragma unroll
      for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) {
          constexpr int TU = Cfg::TU;
          const int slot = (tn / TU == Cfg::NGATE - 1) ? SLOT_LAST * TU + tn % TU : tn;
          MmaOp<T>::run(acc[tm][slot], fa[tm], fb[tn]);
        }
    }
  };
  static_for<D>([&](auto d) __attribute__((always_inline)) { gload(decltype(d)::value, d); });
  sstore(0, IntC<0>{});
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += D) {
    static_for<D>([&](auto d) __attribute__((always_inline)) {
      constexpr int dd = decltype(d)::value;
      const int kt = kt0 + dd;
      if (kt < nk) {                               // block-uniform
        gload(kt + D, d);
        if (kt_begin + kt >= nk1) compute(kt & 1, IntC<SLOT2>{});
        else compute(kt & 1, IntC<SLOT1>{});
        sstore((kt + 1) & 1, IntC<(dd + 1) % D>{});
        __syncthreads();
      }
    });
  }
}

template <class Cfg, int NACC, int SLOT_LAST, class ASrc, class BSrc>
__device__ __forceinline__ void mma_mainloop_nt(f32x16 (&acc)[Cfg::TM][NACC], const ASrc& asrc, const BSrc& bsrc,
                                                int K, char* smem) {
  mma_mainloop_nt<Cfg, NACC, SLOT_LAST>(acc, asrc, bsrc, K, smem, (int)threadIdx.x, 0, (K + Cfg::BK - 1) / Cfg::BK);
}

// ---------------------------------------------------------------------------------------------------
// TN engine: contraction over rows.
template <typename T_, int BM_, int BN_, int BKR_, int WM_, int WN_, int DEPTH_ = 1> struct TileTN {
  using T = T_;
  static constexpr int BM = BM_, BN = BN_, BKR = BKR_, WM = WM_, WN = WN_;
  static constexpr int DEPTH = DEPTH_;         // row tiles kept in flight in registers (global-load prefetch depth)
  static constexpr int NTHREADS = 64 * WM * WN;
  static constexpr int EPC = 16 / sizeof(T);
  static constexpr int PADB = sizeof(T) == 2 ? 64 : 16;
  static constexpr int ROWA = BM * (int)sizeof(T) + PADB, ROWBB = BN * (int)sizeof(T) + PADB;
  static constexpr int WTM = BM / WM, WTN = BN / WN;
  static constexpr int TM = WTM / 32, TN = WTN / 32;
  static constexpr int A_BYTES = BKR * ROWA, B_BYTES = BKR * ROWBB;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int SMEM_BYTES = 2 * STAGE_BYTES;
  static constexpr int ACH = BM / EPC, BCH = BN / EPC;   // chunks per tile row
  static constexpr int A_IT = (BKR * ACH + NTHREADS - 1) / NTHREADS;
  static constexpr int B_IT = (BKR * BCH + NTHREADS - 1) / NTHREADS;
  static_assert(BKR % 16 == 0, "BKR must be a multiple of 16");
  static __device__ __forceinline__ int row_of(int wm, int tm, int reg, int lane) {
    return wm * WTM + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
  }
  static __device__ __forceinline__ int col_of(int wn, int tn, int lane) { return wn * WTN + tn * 32 + (lane & 31); }
};

template <typename T> struct TnFrag;
template <> struct TnFrag<bf16_t> {
  // fragment of a 32-row operand tile for one 16-k step, from an LDS image [k][m] (m contiguous)
  static __device__ __forceinline__ uint4 load(const char* s, int rowbytes, int kbase, int mbase, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int k = kbase + 8 * (g >> 1) + q;
    const int m = mbase + 16 * (g & 1) + 4 * p;
    typedef __attribute__((address_space(3))) s16x4* lptr;
    const char* a0 = s + k * rowbytes + m * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * rowbytes));
    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
  }
  static constexpr int KSTEP = 16;
  static __device__ __forceinline__ void mma(f32x16& c, const uint4& a, const uint4& b) { MmaOp<bf16_t>::run(c, a, b); }
};
template <> struct TnFrag<float> {
  // one f32 per lane: A[m = lane&31][k = lane>>5] for a 2-k step
  static __device__ __forceinline__ uint4 load(const char* s, int rowbytes, int kbase, int mbase, int lane) {
    const float* p = reinterpret_cast<const float*>(s + (kbase + (lane >> 5)) * rowbytes) + mbase + (lane & 31);
    const float* p2 = reinterpret_cast<const float*>(s + (kbase + 2 + (lane >> 5)) * rowbytes) + mbase + (lane & 31);
    return make_uint4(__float_as_uint(*p), __float_as_uint(*p2), 0, 0);
  }
  static constexpr int KSTEP = 4;
  static __device__ __forceinline__ void mma(f32x16& c, const uint4& a, const uint4& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  }
};

// fp32 sum of the EPC elements of one staged 16-byte chunk into cs[] (fused column sums of the TN engine)
template <typename T> __device__ __forceinline__ void chunk_accumulate(const uint4& v, float (&cs)[16 / sizeof(T)]);
template <> __device__ __forceinline__ void chunk_accumulate<bf16_t>(const uint4& v, float (&cs)[8]) {
  const bf16x8 x = __builtin_bit_cast(bf16x8, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] += (float)x[e];
}
template <> __device__ __forceinline__ void chunk_accumulate<float>(const uint4& v, float (&cs)[4]) {
  cs[0] += __uint_as_float(v.x); cs[1] += __uint_as_float(v.y); cs[2] += __uint_as_float(v.z); cs[3] += __uint_as_float(v.w);
}

// acc += A[r0:r1, m0:m0+BM]^T B[r0:r1, n0:n0+BN].  Columns >= Mtot / Ntot and rows >= r1 read as zero.
// csa / csb (block-uniform): also sum this thread's staged chunks of A / B over the rows into ca / cb - the thread always
// stages the same EPC columns (tid % ACH resp. tid % BCH), so the caller gets the operand's column sums for free.
template <class Cfg>
__device__ __forceinline__ void mma_mainloop_tn(f32x16 (&acc)[Cfg::TM][Cfg::TN], const typename Cfg::T* __restrict__ A, long lda,
                                                int m0, int Mtot, const typename Cfg::T* __restrict__ B, long ldb, int n0,
                                                int Ntot, int r0, int r1, char* smem, bool csa, bool csb,
                                                float (&ca)[Cfg::EPC], float (&cb)[Cfg::EPC]) {
  static_assert(Cfg::NTHREADS % Cfg::ACH == 0 && Cfg::NTHREADS % Cfg::BCH == 0, "a thread must keep its columns across chunks");
  using T = typename Cfg::T;
  constexpr int NT = Cfg::NTHREADS, EPC = Cfg::EPC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  constexpr int D = Cfg::DEPTH;
  uint4 ra[D][Cfg::A_IT], rb[D][Cfg::B_IT];   // register ring: tile j lives in slot j % D (static indices only)
  // descriptors start at this block's first row (r0) so that 32-bit byte offsets cover any split of a long operand
  const __amdgpu_buffer_rsrc_t arsrc = pg_make_rsrc(A + (long)r0 * lda);
  const __amdgpu_buffer_rsrc_t brsrc = pg_make_rsrc(B + (long)r0 * ldb);
  const int nk = (r1 - r0 + Cfg::BKR - 1) / Cfg::BKR;
  auto gload = [&](int kt, auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    const int rr = r0 + kt * Cfg::BKR;
#pragma unroll
    for (int i = 0; i < Cfg::A_IT; ++i) {
      int id = tid + i * NT, kr = id / Cfg::ACH, c = id % Cfg::ACH;
      bool ok = id < Cfg::BKR * Cfg::ACH && rr + kr < r1 && m0 + c * EPC < Mtot;
      ra[slot][i] = pg_buf_load16(arsrc, ok ? (unsigned)(((long)(rr + kr - r0) * lda + m0 + c * EPC) * (long)sizeof(T)) : PG_OOB, 0);
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_IT; ++i) {
      int id = tid + i * NT, kr = id / Cfg::BCH, c = id % Cfg::BCH;
      bool ok = id < Cfg::BKR * Cfg::BCH && rr + kr < r1 && n0 + c * EPC < Ntot;
      rb[slot][i] = pg_buf_load16(brsrc, ok ? (unsigned)(((long)(rr + kr - r0) * ldb + n0 + c * EPC) * (long)sizeof(T)) : PG_OOB, 0);
    }
  };
  auto sstore = [&](int buf, auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    char* sa = smem + buf * Cfg::STAGE_BYTES;
    char* sb = sa + Cfg::A_BYTES;
#pragma unroll
    for (int i = 0; i < Cfg::A_IT; ++i) {
      int id = tid + i * NT, kr = id / Cfg::ACH, c = id % Cfg::ACH;
      if (id < Cfg::BKR * Cfg::ACH) *reinterpret_cast<uint4*>(sa + kr * Cfg::ROWA + c * 16) = ra[slot][i];
      if (csa && id < Cfg::BKR * Cfg::ACH) chunk_accumulate<T>(ra[slot][i], ca);
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_IT; ++i) {
      int id = tid + i * NT, kr = id / Cfg::BCH, c = id % Cfg::BCH;
      if (id < Cfg::BKR * Cfg::BCH) *reinterpret_cast<uint4*>(sb + kr * Cfg::ROWBB + c * 16) = rb[slot][i];
      if (csb && id < Cfg::BKR * Cfg::BCH) chunk_accumulate<T>(rb[slot][i], cb);
    }
  };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* sa = smem + buf * Cfg::STAGE_BYTES;
    const char* sb = sa + Cfg::A_BYTES;
    // fragments of k-step s+1 are requested BEFORE the MFMAs of step s (two register sets, pinned by scheduling
    // barriers): left alone hipcc sinks every ds_read next to its MFMA and each k-step pays a full LDS latency - with
    // one or two waves per SIMD nothing hides it (the kernel sat at 24 % matrix-pipe utilisation, LDS 24 % busy,
    // L2 25 % busy: latency-bound inside the wave)
    constexpr int S = Cfg::BKR / TnFrag<T>::KSTEP;
    uint4 fa[2][Cfg::TM], fb[2][Cfg::TN];
    auto lfrag = [&](int s, int set) __attribute__((always_inline)) {
#pragma unroll
      for (int tm = 0; tm < Cfg::TM; ++tm)
        fa[set][tm] = TnFrag<T>::load(sa, Cfg::ROWA, s * TnFrag<T>::KSTEP, wm * Cfg::WTM + tm * 32, lane);
#pragma unroll
      for (int tn = 0; tn < Cfg::TN; ++tn)
        fb[set][tn] = TnFrag<T>::load(sb, Cfg::ROWBB, s * TnFrag<T>::KSTEP, wn * Cfg::WTN + tn * 32, lane);
    };
    lfrag(0, 0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
      if (s + 1 < S) lfrag(s + 1, (s + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) TnFrag<T>::mma(acc[tm][tn], fa[s & 1][tm], fb[s & 1][tn]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // tiles 0..D-1 in flight, tile 0 staged; then per tile: refill the slot staged one iteration ago with tile kt+D (past
  // the end it is fully predicated off: zeros, no traffic), contract tile kt, stage tile kt+1 (the oldest loads in flight).
  // The tile count is rounded UP to a multiple of D and the unrolled body has no conditional: with a branch per tile hipcc
  // loses the vmcnt bookkeeping across the back edge and drains the whole ring (vmcnt(0)) once per D tiles; the padding
  // tiles contract zeros
  static_for<D>([&](auto d) __attribute__((always_inline)) { gload(decltype(d)::value, d); });
  sstore(0, IntC<0>{});
  __syncthreads();
  const int nk_pad = (nk + D - 1) / D * D;
  for (int kt0 = 0; kt0 < nk_pad; kt0 += D) {
    static_for<D>([&](auto d) __attribute__((always_inline)) {
      constexpr int dd = decltype(d)::value;
      const int kt = kt0 + dd;
      gload(kt + D, d);
      compute(kt & 1);
      sstore((kt + 1) & 1, IntC<(dd + 1) % D>{});
      __syncthreads();
    });
  }
}
