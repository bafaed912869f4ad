// Group-persistent decoder recurrence, forward, TWO waves per SIMD (round 3).  Same grouping, inter-workgroup protocol,
// operand layout and arithmetic as pg_persist_dec.hip (read that file first); what changes is how a workgroup spends its CU.
//
// Why: with one wave per SIMD a 16-row tile is a chain - LDS-DMA issue, 96 MFMAs, gate epilogue, stores - on ONE
// instruction stream, and the matrix pipe idles through everything that is not an MFMA (2.0 us per tile against 0.64 us of
// MFMA time; tools/mfma_rate.hip: 16 cycles per v_mfma_f32_16x16x32_bf16 whatever the shape, and only ~2 plain VALU
// instructions hide in one's shadow).  Two waves per SIMD overlap the phases in hardware, but each wave then owns 256
// registers and the workgroup's stationary weights alone are 384 registers' worth per SIMD lane.  The split that fits:
//
//   waves 0-3 ("h side", role 0): W_hh rows of the wave's 16 units, all gates (r, z: 128 AGPRs; n: 64 VGPRs);
//   waves 4-7 ("x side", role 1): W_ih rows of the same units (r, z: 128 AGPRs; n: K 0-127 in 16 VGPRs, K 128-511 in LDS).
//
// Waves w and w + 4 form a PAIR on one SIMD (a workgroup's waves are dealt round-robin over the four SIMDs).  Each runs
// the 48 MFMAs of its K half over a 16-row tile, and the pair splits the gate epilogue BY UNITS: a lane owns 4 units of
// one sequence; role 0 finishes units 0-1, role 1 units 2-3, each handing the other's half of its partial accumulators
// over through LDS (3 gates x 2 floats per lane and direction).  A first version that left the whole epilogue to the x side
// measured SLOWER than the 4-wave kernel (960 against 860 us: the x wave's MFMA phase and epilogue ran back to back while
// the h wave idled - knock-outs: x MFMAs 220 us, epilogue 310 us, h MFMAs 40 us).  With the epilogue split, the two waves
// of a pair are STAGGERED by half an iteration, so that one's MFMA phase runs beside the other's VALU / store phase:
//
//   iteration i           role 0 (h)                                      role 1 (x)
//   (Bar(i-1) passed)     issue LDS-DMA of tile i+1 -> buf (i+1)&1        row-segment + relayed stores of tile i-2
//                         epilogue(i-1), units 0-1: own h partials        48 MFMAs over x of tile i
//                           + x partials from P[(i-1)&1]                  units 0-1 of the result -> P[i&1] (x->h)
//                         48 MFMAs over h_{t-1} of tile i                 epilogue(i-1), units 2-3: own x partials (kept in
//                         units 2-3 of the result -> P[i&1] (h->x)          registers across the MFMAs) + h partials from P[(i-1)&1]
//                         state slice of tile i -> registers              state slice of tile i -> registers
//                         wait: DMA(i+1) landed;  Bar(i)                  Bar(i)
//   (the x side has the registers for the second set of partial sums: its n-gate weights sit half in LDS; the h side, which
//   keeps three quarters of W_hn in VGPRs, runs its epilogue first and so never holds two tiles' sums)
//
// LDS hazards, each separated by a Bar: buf (i+1)&1 was last read in iteration i-1; P[i&1] is written in iteration i, read
// in i+1, rewritten in i+2; the staging block S[k&1] is written by epilogue(k) in iteration k+1, read in k+2, rewritten in
// k+3.  Role 0 issues only loads (the LDS-DMA, from asm: hipcc adds no waits of its own), role 1 only stores, so neither
// ever waits for the other kind (loads and stores of a wave share ONE in-order counter, and a store takes 2-3 us to be
// acknowledged): the two stores role 0's epilogue owes (state, lane-order saves) are RELAYED through the x -> h half of
// P[k&1], which role 0 has just consumed, and issued by its partner an iteration later; role 1 drains its stores once per
// step, before the arrival on the (group, layer) counter.
// Layer 0 has no x side in the recurrence (its input projection is the gather table): its role-1 waves run their half of
// the epilogue only.
//
// The saved gate activations leave in a pair-split lane order (sv_lane = 2: role 0 stores {r, z, n, hn} of units 0-1 as
// 16 bytes per lane into the (r, z) region, role 1 those of units 2-3 into the (n, hn) region); the BPTT kernel reads them
// back lane for lane (pg_persist_bwd.hip, SVL).  Only the training configuration (dropout between layers, lane-order saves:
// the FAST case of pg_persist_dec.hip) takes this kernel; every other case keeps the 4-wave kernel.
#include <hip/hip_runtime.h>
#include "pg_mma.h"
#include "pg_persist.h"

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

namespace {

constexpr int SYNC_STRIDE = 16;
constexpr int SYNC_ERR = PG_MAX_LAYERS * PG_PERSIST_MAX_GROUPS * SYNC_STRIDE;
constexpr int AUX_SC1 = 16;

// weights of gates r, z in AGPRs, of gate n in VGPRs (a register-resident fragment or one just read from LDS)
template <bool NOP>
__device__ __forceinline__ void mma12_aav(f32x4& c0, f32x4& c1, f32x4& c2, const bf16x8* wa, const bf16x8* wb, const bf16x8* wc,
                                          const bf16x8* x) {
  if constexpr (NOP) asm("s_nop 1\n\t" PG_MMA12_BODY PG_MMA12_OPS("a", "a", "v"));
  else asm(PG_MMA12_BODY PG_MMA12_OPS("a", "a", "v"));
}
// Two K steps x three gates (6 MFMAs): the fragment registers of a group halve (8 + 8 instead of 16 + 16), which is what
// lets both roles run their tile loops without a spill at 128 VGPRs; with two waves per SIMD the shorter groups cost nothing.
#define PG_MMA6_BODY                                               \
  "v_mfma_f32_16x16x32_bf16 %0, %3, %9, %0\n\ts_nop 0\n\t"       \
  "v_mfma_f32_16x16x32_bf16 %1, %5, %9, %1\n\ts_nop 0\n\t"       \
  "v_mfma_f32_16x16x32_bf16 %2, %7, %9, %2\n\ts_nop 0\n\t"       \
  "v_mfma_f32_16x16x32_bf16 %0, %4, %10, %0\n\ts_nop 0\n\t"      \
  "v_mfma_f32_16x16x32_bf16 %1, %6, %10, %1\n\ts_nop 0\n\t"      \
  "v_mfma_f32_16x16x32_bf16 %2, %8, %10, %2\n\ts_nop 0"
#define PG_MMA6_OPS : "+v"(c0), "+v"(c1), "+v"(c2) : "a"(wa[0]), "a"(wa[1]), "a"(wb[0]), "a"(wb[1]), "v"(wc[0]), "v"(wc[1]), "v"(x[0]), "v"(x[1])
template <bool NOP>
__device__ __forceinline__ void mma6_aav(f32x4& c0, f32x4& c1, f32x4& c2, const bf16x8* wa, const bf16x8* wb, const bf16x8* wc,
                                         const bf16x8* x) {
  if constexpr (NOP) asm("s_nop 1\n\t" PG_MMA6_BODY PG_MMA6_OPS);
  else asm(PG_MMA6_BODY PG_MMA6_OPS);
}
// matrix-pipe write -> VALU read wait states; tied to the accumulators so that every MFMA asm is ordered before it
#define PG_MFMA_DRAIN3(a0, a1, a2) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2))

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  bf16x2 o;
  o[0] = (bf16_t)a; o[1] = (bf16_t)b;
  return __builtin_bit_cast(unsigned, o);
}

// LDS-DMA from asm (invisible to hipcc): 16 bytes per lane from a per-lane 64-bit address, lane-linear at the wave-uniform
// LDS byte address m.  sc1: rows another workgroup published during this launch (served by L2, never by the CU's L1).
__device__ __forceinline__ void dma16_sc1(const void* src, unsigned m) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(src), "s"(m) : "memory", "m0");
}
__device__ __forceinline__ void dma16(const void* src, unsigned m) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m) : "memory", "m0");
}
// the same from a wave-uniform base (SGPR pair) + a 32-bit lane offset: no 64-bit address arithmetic per lane
__device__ __forceinline__ void dma16s_sc1(const void* sbase, unsigned voff, unsigned m) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(voff), "s"(sbase), "s"(m) : "memory", "m0");
}
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned m) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m) : "memory", "m0");
}

__device__ __forceinline__ bool wait_ge8(unsigned* p, unsigned target, unsigned* err, unsigned* sticky, unsigned limit) {
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if ((spins & 63u) == 63u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    if (spins > limit) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int V> struct IC8 { static constexpr int value = V; };

template <int H> struct Geo8 {
  static constexpr int KS = H / 32;                   // MFMA K steps over K = H
  static constexpr int NCU = H / 64;
  // n-gate weights that do not fit the registers live in LDS: the upper 6 of 16 K steps of W_in (x side) and of W_hn (h
  // side) - with 8 / 4 or 4 / 8 the side with 4 spilled ~10 registers in its tile loop.  One unit's LDS row holds KSL * 4
  // chunks of 16 bytes = 384 bytes, so consecutive rows alternate between the two halves of the 64 banks and chunk q of row
  // u sits at position q ^ ((u >> 1) & 7): the 16 rows a ds_read_b128 lane group touches land in 16 different bank groups
  static constexpr int KSL_X = 6, KSL_H = 6;
  static_assert((KSL_X * 64) % 256 == 128 && (KSL_H * 64) % 256 == 128 && KSL_X % 2 == 0 && KSL_H % 2 == 0, "LDS weight swizzle");
  static constexpr int WX_BYTES = 64 * KSL_X * 64;    // 24 KB
  static constexpr int WH_BYTES = 64 * KSL_H * 64;    // 24 KB
  static constexpr int WIN_BYTES = WX_BYTES + WH_BYTES;
  static constexpr int ACT_BYTES = 16 * 2 * H * 2;    // tile rows [h_{t-1} | x] (layer 0: half)
  static constexpr int CARRY_BYTES = 16 * 64 * 4;
  static constexpr int GATE_BYTES = 3 * CARRY_BYTES;
  static constexpr int TILE_BYTES = ACT_BYTES + CARRY_BYTES;
  static constexpr int TILE0_BYTES = ACT_BYTES / 2 + CARRY_BYTES + 2 * GATE_BYTES;
  static constexpr int TOK_BYTES = 128 * 4;
  static constexpr int P_DIR = 3 * 512;               // one direction of one pair: [gate 3][lane 64] f32x2
  static constexpr int P_PAIR = 2 * P_DIR;            // h -> x, then x -> h
  static constexpr int P_BUF = 4 * P_PAIR;            // 12 KB
  static constexpr int STG_ROW = 144, STG_ARR = 16 * STG_ROW, STG_BUF = 3 * STG_ARR;
  static constexpr int FRONT0 = 2 * TILE0_BYTES + TOK_BYTES + WH_BYTES;
  static constexpr int FRONT = (WIN_BYTES + 2 * TILE_BYTES) > FRONT0 ? (WIN_BYTES + 2 * TILE_BYTES) : FRONT0;
  static constexpr int P_OFF = FRONT;
  static constexpr int STG_OFF = P_OFF + 2 * P_BUF;
  static constexpr int BIAS_OFF = STG_OFF + 2 * STG_BUF;   // [b_hn | b_ir+b_hr | b_iz+b_hz | b_in] of the slice's 64 units, fp32
  static constexpr int SMEM = BIAS_OFF + 4 * 64 * 4 + 16;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
};

// LK: 0 = layer 0, 1 = a middle layer, 2 = the top layer (no dropout output).  ROLE: 0 = h side, 1 = x side.
// KO: development knock-outs (timing only, results invalid; library built with -DPG_FWD8_KNOCKOUTS): 1 no LDS-DMA in the
// tile loop, 2 no MFMAs, 4 no epilogue, 8 no row-segment stores.
template <int H, int LK, int ROLE, int KO>
__device__ __forceinline__ void fwd8_wave(const DecFwdPersist& a, const Sched& sd, int g, int l, int c, char* smem, int Tg) {
  using G = Geo8<H>;
  constexpr bool L0 = LK == 0;
  constexpr bool has_drop = LK != 2;
  constexpr bool HS = ROLE == 0;                      // the h side
  constexpr bool MMA_ = ROLE == 0 || LK != 0;
  constexpr bool MMA = MMA_;                          // this wave runs MFMAs
  constexpr int KS = G::KS, NCU = G::NCU;
  constexpr int KSL = !MMA_ ? 0 : (ROLE == 0 ? G::KSL_H : G::KSL_X), WROW = KSL * 64;
  constexpr int PARTS = L0 ? 1 : 2;
  constexpr int TILEB = L0 ? G::TILE0_BYTES : G::TILE_BYTES;
  constexpr int ACTB = G::ACT_BYTES / (L0 ? 2 : 1);
  constexpr int narr = has_drop ? 3 : 2;              // row-major bf16 outputs: h, [dropout(h)], h_prev
  constexpr int KREG = KS - KSL;                      // n-gate K steps held in VGPRs
  constexpr int UH = 2 * ROLE;                        // first of this wave's 2 units among the lane's 4
  constexpr bool PRIO = !(KO & 16);
  // KO & 32: cycle stamps (s_memtime) of pair 0 of workgroup (group 0, layer 1, slice 0) at step 20, 5 per iteration and role
#define PG8_STAMP(slot)                                                                                         \
  do {                                                                                                          \
    if constexpr ((KO & 32) != 0) {                                                                             \
      if (a.stamps && g == 0 && l == 1 && c == 0 && wave == 0 && t == 20 && i < 8 && lane == 0)                 \
        a.stamps[8 + (ROLE * 8 + i) * 8 + (slot)] = __builtin_amdgcn_s_memtime();                               \
    }                                                                                                           \
  } while (0)
  // step-level stamps of the same pair, steps 20 .. 23: slots 0 .. 6 behind the tile stamps
#define PG8_STEP_STAMP(slot)                                                                                    \
  do {                                                                                                          \
    if constexpr ((KO & 32) != 0) {                                                                             \
      if (a.stamps && g == 0 && l == 1 && c == 0 && wave == 0 && t >= 20 && t < 24 && lane == 0)                \
        a.stamps[8 + 128 + (ROLE * 4 + (t - 20)) * 8 + (slot)] = __builtin_amdgcn_s_memtime();                  \
    }                                                                                                           \
  } while (0)
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) - 4 * ROLE;   // pair index 0..3
  const int lr = lane & 15, lq = lane >> 4, hi = lr >> 2;
  const int NG = a.NG;
  const int uw = c * 64 + wave * 16;
  const int u0 = uw + lq * 4;                         // the lane's 4 units; this wave finishes u0 + UH, u0 + UH + 1
  // layer 0 keeps its (h-side) LDS weights behind the token block; layers >= 1 in front of the tile buffers
  char* sWin = L0 ? smem + 2 * G::TILE0_BYTES + G::TOK_BYTES : smem + (HS ? G::WX_BYTES : 0);
  char* sT = L0 ? smem : smem + G::WIN_BYTES;
  int* sTok = reinterpret_cast<int*>(smem + 2 * G::TILE0_BYTES);   // layer 0 only
  volatile unsigned* sFlag = reinterpret_cast<volatile unsigned*>(smem + G::SMEM - 16);
  const unsigned ldsT = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)sT);
  char* sPout = smem + G::P_OFF + wave * G::P_PAIR + ROLE * G::P_DIR + lane * 8;         // what this wave hands over
  const char* sPin = smem + G::P_OFF + wave * G::P_PAIR + (1 - ROLE) * G::P_DIR + lane * 8;   // what its partner hands over
  char* sStage = smem + G::STG_OFF;
  float* sBias = reinterpret_cast<float*>(smem + G::BIAS_OFF);

  // ---- stationary weights: rows of W_hh (h side) / W_ih (x side, layers >= 1) of the wave's 16 units
  const bf16_t* W = HS ? a.Whh[l] : a.Wih[l];
  bf16x8 wr[MMA ? KS : 1], wz[MMA ? KS : 1], wn[MMA ? KREG : 1];
  if constexpr (MMA) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wr[ks] = *reinterpret_cast<const bf16x8*>(W + (long)(0 * H + uw + lr) * H + (ks * 4 + lq) * 8);
      wz[ks] = *reinterpret_cast<const bf16x8*>(W + (long)(1 * H + uw + lr) * H + (ks * 4 + lq) * 8);
    }
#pragma unroll
    for (int ks = 0; ks < KREG; ++ks)
      wn[ks] = *reinterpret_cast<const bf16x8*>(W + (long)(2 * H + uw + lr) * H + (ks * 4 + lq) * 8);
  }
  if constexpr (MMA) {   // LDS image of the upper K range of the slice's 64 n-gate rows (swizzle: Geo8)
    for (int id = tid - 256 * ROLE; id < 64 * (KSL * 4); id += 256) {
      const int u = id / (KSL * 4), ch = id % (KSL * 4);
      *reinterpret_cast<uint4*>(sWin + u * WROW + ((ch ^ ((u >> 1) & 7)) << 4)) =
          *reinterpret_cast<const uint4*>(W + (long)(2 * H + c * 64 + u) * H + KREG * 32 + ch * 8);
    }
  }
  if constexpr (!HS) {
    if (tid - 256 < 64) {   // the biases are re-read from LDS in every epilogue (registers neither side has)
      const int u = tid - 256;
      sBias[u] = a.bhn[l][c * 64 + u];
#pragma unroll
      for (int gt = 0; gt < 3; ++gt) sBias[(1 + gt) * 64 + u] = L0 ? 0.0f : a.bias_gi[l][gt * H + c * 64 + u];
    }
  }
  const float* sB = sBias + wave * 16 + lq * 4 + UH;
  const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  bf16_t* hseq = a.hseq[l];
  const bf16_t* xsrc = L0 ? nullptr : a.hdrop[l - 1];
  float* carry = a.carry[l];
  const __amdgpu_buffer_rsrc_t r_hseq = pg_make_rsrc(hseq);
  const __amdgpu_buffer_rsrc_t r_hdrop = pg_make_rsrc(has_drop ? a.hdrop[l] : hseq);
  const __amdgpu_buffer_rsrc_t r_hprev = pg_make_rsrc(a.sv[l][4]);
  const __amdgpu_buffer_rsrc_t r_carry = pg_make_rsrc(carry);
  const __amdgpu_buffer_rsrc_t r_sv0 = pg_make_rsrc(a.sv[l][0]);   // lane-order saves of units 0-1 (h side, relayed)
  const __amdgpu_buffer_rsrc_t r_sv2 = pg_make_rsrc(a.sv[l][2]);   // ... of units 2-3 (x side)
  unsigned* cnt_own = a.sync + (g * PG_MAX_LAYERS + l) * SYNC_STRIDE;
  unsigned* cnt_low = a.sync + (g * PG_MAX_LAYERS + (L0 ? 0 : l - 1)) * SYNC_STRIDE;
  unsigned* err = a.sync + SYNC_ERR;
  __syncthreads();

  for (int t = 0; t < Tg; ++t) {
    PG8_STEP_STAMP(0);
    if constexpr (HS) {
      if (wave == 0) {   // (wave-uniform) own layer's step t-1 and the lower layer's step t, both counters polled at once
        const bool ok = pg_wait_ge2(cnt_own, (unsigned)(NCU * t), cnt_low, (unsigned)(NCU * (t + 1)), !L0, err, a.fail, a.spin_limit);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) *sFlag = ok ? 1u : 0u;
      }
    }
    const int o = sd.off[t], Bt = sd.off[t + 1] - o;
    const int op = t > 0 ? sd.off[t - 1] : 0;
    const int nlive = (Bt - g + NG - 1) / NG;
    const int ntile = (nlive + 15) >> 4;
    if constexpr (L0 && HS) {
      if (tid < 128) sTok[tid] = a.tok[o + (tid < nlive ? g + NG * tid : g)];
    }
    __syncthreads();                                                               // Bar S
    if (*sFlag == 0u) return;
    PG8_STEP_STAMP(1);
    const bf16_t* hprev_base = t == 0 ? a.h0_op : hseq + (long)op * H;
    const bf16_t* x_base = L0 ? hprev_base : xsrc + (long)o * H;
    const float* carry_src = (t == 0 ? a.h0_32 : carry) + c * 64;

    // (h side) one tile's LDS-DMA, this wave's share: 4 rows x PARTS row halves (1 KB each, XOR-swizzled on the source
    // side) + a quarter of the epilogue inputs (state slice; layer 0: gather-table and z-projection rows)
    auto stage = [&](int j, int buf) __attribute__((always_inline)) {
      const unsigned tb = ldsT + buf * TILEB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // the row and its sequence are wave-uniform: scalar base, the lane adds only its swizzled chunk
        const int row = wave * 4 + i;
        const int b = g + NG * (16 * j + row);
        const int bs = b < Bt ? b : g;
        const unsigned vo = (unsigned)((lane ^ row) * 16);
        dma16s_sc1(hprev_base + (long)bs * H, vo, tb + row * (PARTS * 1024));
        if constexpr (!L0) dma16s_sc1(x_base + (long)bs * H, vo, tb + row * (PARTS * 1024) + 1024);
      }
      {
        const int row = wave * 4 + (lane >> 4), ch = (lane & 15) ^ row;
        const int i = 16 * j + row;
        const int b = g + NG * i;
        const int bs = b < Bt ? b : g;
        const unsigned eb = tb + ACTB + wave * 1024;
        const unsigned vo = (unsigned)(bs * H + ch * 4) * 4u;          // < 2^31: B * H * 4 bytes
        dma16s(carry_src, vo, eb);
        if constexpr (L0) {
          const unsigned vt = (unsigned)(sTok[i < 128 ? i : 0] * 3 * H + c * 64 + ch * 4) * 4u;
          const unsigned vz = (unsigned)(bs * 3 * H + c * 64 + ch * 4) * 4u;
#pragma unroll
          for (int gt = 0; gt < 3; ++gt) {
            dma16s(a.table + gt * H, vt, eb + G::CARRY_BYTES + gt * G::CARRY_BYTES);
            dma16s(a.zproj + gt * H, vz, eb + G::CARRY_BYTES + G::GATE_BYTES + gt * G::CARRY_BYTES);
          }
        }
      }
    };

    f32x2 own[3];             // this wave's partial sums (its K half) of ITS units of the tile whose epilogue is due
    f32x2 hp2, g2[3];         // the same tile's previous state and (layer 0) table + z-projection rows, this wave's units
    // 48 MFMAs of tile i over this wave's K half; the partner's units go to P[i & 1], the wave's own stay in `own`
    auto mfma_tile = [&](int i) __attribute__((always_inline)) {
      if constexpr (MMA) {
        f32x4 acc[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!(KO & 2)) {
          const char* A = sT + (i & 1) * TILEB + lr * (PARTS * 1024) + (HS ? 0 : 1024) + ((lq ^ lr) & 3) * 16;
          const int sw = (lr >> 1) & 7;
          const char* Wn = sWin + (wave * 16 + lr) * WROW + ((lq ^ sw) & 3) * 16;
          // groups of 2 K steps, the fragments of group gk + 1 requested before the MFMAs of group gk are issued: without
          // the read-ahead every group exposed an LDS round trip (8 x ~300 cycles per tile and wave, measured as 0.6 us)
          bf16x8 f[KS / 2][2], wl[KS / 2][2];
          // (the x side, 8 registers tighter, reads its LDS weight fragments with the group that uses them)
          constexpr bool WAHEAD = HS;
          auto fetch = [&](auto GKc) __attribute__((always_inline)) {
            constexpr int gk = decltype(GKc)::value;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int ks = gk * 2 + k;
              f[gk][k] = *reinterpret_cast<const bf16x8*>(A + ((ks ^ hi) << 6));
              if (WAHEAD && ks >= KREG) wl[gk][k] = *reinterpret_cast<const bf16x8*>(Wn + (((ks - KREG) ^ (sw >> 2)) << 6));
            }
          };
          constexpr int AHEAD = 1;               // groups requested ahead of the one whose MFMAs are issued (2 on the h side: spills)
          fetch(IC8<0>{});
          if constexpr (AHEAD > 1) fetch(IC8<1>{});
          __builtin_amdgcn_sched_barrier(0);
          // the wave in its MFMA phase takes precedence at the SIMD's issue port: an MFMA needs one slot in 16 cycles, the
          // partner's epilogue VALU fills the rest; without it the older wave's VALU starves the younger wave's MFMAs
          if (PRIO) __builtin_amdgcn_s_setprio(2);
          static_for<KS / 2>([&](auto GKc) __attribute__((always_inline)) {
            constexpr int gk = decltype(GKc)::value;
            if constexpr (gk + AHEAD < KS / 2) fetch(IC8<gk + AHEAD>{});
            if constexpr (!WAHEAD) {
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                const int ks = gk * 2 + k;
                if (ks >= KREG) wl[gk][k] = *reinterpret_cast<const bf16x8*>(Wn + (((ks - KREG) ^ (sw >> 2)) << 6));
              }
            }
            bf16x8 wsel[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int ks = gk * 2 + k;
              wsel[k] = ks >= KREG ? wl[gk][k] : wn[ks < KREG ? ks : 0];
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (gk == 0) mma6_aav<true>(acc[0], acc[1], acc[2], &wr[gk * 2], &wz[gk * 2], wsel, f[gk]);
            else mma6_aav<false>(acc[0], acc[1], acc[2], &wr[gk * 2], &wz[gk * 2], wsel, f[gk]);
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        PG_MFMA_DRAIN3(acc[0], acc[1], acc[2]);
        char* P = sPout + (i & 1) * G::P_BUF;
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
          *reinterpret_cast<f32x2*>(P + gt * 512) = f32x2{acc[gt][2 - UH], acc[gt][3 - UH]};   // the partner's units
          own[gt] = f32x2{acc[gt][UH], acc[gt][UH + 1]};
        }
      }
    };
    // epilogue inputs of tile i from its tile buffer -> registers (the buffer is refilled after the next barrier)
    auto load_epi = [&](int i) __attribute__((always_inline)) {
      const char* E = sT + (i & 1) * TILEB + ACTB + lr * 256 + (((wave * 4 + lq) ^ lr) << 4) + UH * 4;
      hp2 = *reinterpret_cast<const f32x2*>(E);
      if constexpr (L0) {
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
          g2[gt] = *reinterpret_cast<const f32x2*>(E + G::CARRY_BYTES + gt * G::CARRY_BYTES) +
                   *reinterpret_cast<const f32x2*>(E + G::CARRY_BYTES + G::GATE_BYTES + gt * G::CARRY_BYTES);
      }
    };
    // (x side) row-segment stores of the rows staged for tile j (all eight waves' units): 128 lanes per array, 16 B per lane
    auto row_stores = [&](int j) __attribute__((always_inline)) {
      if constexpr (!HS && !(KO & 8)) {
        const char* SB = sStage + (j & 1) * G::STG_BUF;
        const int row = (lane >> 3) + ((wave & 1) << 3), seg = lane & 7;
        const int brow = g + NG * (16 * j + row);
        const unsigned oob = 0x80000000u;
        const unsigned orow = brow < Bt ? (unsigned)((((long)(o + brow) * H + c * 64) * 2) + seg * 16) : oob;
#pragma unroll
        for (int rd = 0; rd < (narr > 2 ? 2 : 1); ++rd) {
          const int arr = 2 * rd + (wave >> 1);                // wave-uniform
          const u32x4 v = *reinterpret_cast<const u32x4*>(SB + (arr < 3 ? arr : 0) * G::STG_ARR + row * G::STG_ROW + seg * 16);
          const unsigned off = arr < narr ? orow : oob;
          if (arr == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r_hseq, (int)off, 0, AUX_SC1);
          else if (arr == 1 && has_drop) __builtin_amdgcn_raw_buffer_store_b128(v, r_hdrop, (int)off, 0, AUX_SC1);
          else __builtin_amdgcn_raw_buffer_store_b128(v, r_hprev, (int)off, 0, 0);
        }
      }
    };
    // (x side) the two stores the h side's epilogue of tile j left in the x -> h half of P[j & 1]: state and lane-order saves
    // of units 0-1 (same lane mapping in both waves of a pair)
    auto relay_stores = [&](int j) __attribute__((always_inline)) {
      if constexpr (!HS && !(KO & 4)) {
        const char* R = smem + G::P_OFF + (j & 1) * G::P_BUF + wave * G::P_PAIR + G::P_DIR;
        const f32x2 hv = *reinterpret_cast<const f32x2*>(R + lane * 8);
        const u32x4 sv4 = *reinterpret_cast<const u32x4*>(R + 512 + lane * 16);
        const int b = g + NG * (16 * j + lr);
        const bool live = b < Bt;
        const unsigned oob = 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), r_carry,
                                              (int)(live ? (unsigned)(((long)b * H + u0) * 4) : oob), 0, 0);
        const unsigned o16 = live ? (unsigned)(((sv_tile_slot(o, t, NG, j, g) * NCU + c) * 256 + wave * 64 + lane) * 16) : oob;
        __builtin_amdgcn_raw_buffer_store_b128(sv4, r_sv0, (int)o16, 0, 0);
      }
    };
    // gate epilogue of tile j for this wave's 2 units: own partials + the partner's (P[j & 1]) -> h; the state and the
    // lane-order saves are stored (x side) or relayed (h side), the row-major outputs staged in S[j & 1]
    auto epilogue = [&](int j) __attribute__((always_inline)) {
      if constexpr (!(KO & 4)) {
        const int b = g + NG * (16 * j + lr);
        const bool live = b < Bt;
        const char* P = sPin + (j & 1) * G::P_BUF;
        f32x2 hpart[3], xpart[3];       // h-side / x-side sums of the gates r, z, n (x side incl. the input biases)
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
          const f32x2 other = (HS && L0) ? f32x2{0.f, 0.f} : *reinterpret_cast<const f32x2*>(P + gt * 512);
          const f32x2 bg = L0 ? g2[gt] : *reinterpret_cast<const f32x2*>(sB + (1 + gt) * 64);
          if constexpr (HS) { hpart[gt] = own[gt]; xpart[gt] = L0 ? bg : other + bg; }
          else { hpart[gt] = other; xpart[gt] = L0 ? bg : own[gt] + bg; }
        }
        const f32x2 bh = *reinterpret_cast<const f32x2*>(sB);
        float r[2], z[2], n[2], hn[2], h[2], hd[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          r[e] = sigmoidf_(xpart[0][e] + hpart[0][e]);
          z[e] = sigmoidf_(xpart[1][e] + hpart[1][e]);
          hn[e] = hpart[2][e] + bh[e];
          n[e] = tanhf_(xpart[2][e] + r[e] * hn[e]);
          h[e] = (1.0f - z[e]) * n[e] + z[e] * hp2[e];
        }
        {
          char* S = sStage + (j & 1) * G::STG_BUF + lr * G::STG_ROW + (wave * 16 + lq * 4 + UH) * 2;
          *reinterpret_cast<unsigned*>(S) = pack_bf16x2(h[0], h[1]);
          int k = 1;
          if constexpr (has_drop) {
            float m[4];
            dropout_scale4(a.seed, l, t, b * a.b_mul + a.b_add, u0 >> 2, a.p_drop, inv_keep, m);
            hd[0] = h[0] * m[UH]; hd[1] = h[1] * m[UH + 1];
            *reinterpret_cast<unsigned*>(S + k * G::STG_ARR) = pack_bf16x2(hd[0], hd[1]);
            ++k;
          }
          *reinterpret_cast<unsigned*>(S + k * G::STG_ARR) = pack_bf16x2(hp2[0], hp2[1]);
        }
        const u32x4 sv4 = u32x4{pack_bf16x2(r[0], r[1]), pack_bf16x2(z[0], z[1]), pack_bf16x2(n[0], n[1]), pack_bf16x2(hn[0], hn[1])};
        if constexpr (HS) {   // relayed: into the x -> h half of P[j & 1], which this wave has just read (see relay_stores)
          char* R = smem + G::P_OFF + (j & 1) * G::P_BUF + wave * G::P_PAIR + G::P_DIR;
          *reinterpret_cast<f32x2*>(R + lane * 8) = f32x2{h[0], h[1]};
          *reinterpret_cast<u32x4*>(R + 512 + lane * 16) = sv4;
        } else {
          const unsigned oob = 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{h[0], h[1]}), r_carry,
                                                (int)(live ? (unsigned)(((long)b * H + u0 + UH) * 4) : oob), 0, 0);
          const unsigned o16 = live ? (unsigned)(((sv_tile_slot(o, t, NG, j, g) * NCU + c) * 256 + wave * 64 + lane) * 16) : oob;
          __builtin_amdgcn_raw_buffer_store_b128(sv4, r_sv2, (int)o16, 0, 0);
        }
      }
    };

    if constexpr (HS) {
      stage(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                                  // Bar P: tile 0 has landed
    asm volatile("" ::: "memory");
    PG8_STEP_STAMP(2);
    for (int i = 0; i < ntile; ++i) {
      PG8_STAMP(0);
      if constexpr (HS) {
        if (!(KO & 1) && i + 1 < ntile) stage(i + 1, (i + 1) & 1);                 // block-uniform
        PG8_STAMP(1);
        if (i >= 1) epilogue(i - 1);
        PG8_STAMP(2);
        mfma_tile(i);
        PG8_STAMP(3);
        load_epi(i);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               // DMA(i+1) has landed (this side issues no stores)
        PG8_STAMP(4);
      } else {
        if (i >= 2) relay_stores(i - 2);                                           // (before this wave rewrites P[i & 1])
        PG8_STAMP(1);
        f32x2 prev[3] = {own[0], own[1], own[2]};
        mfma_tile(i);                                                              // first: the h side's MFMAs come half an iteration later
        PG8_STAMP(2);
        f32x2 cur[3] = {own[0], own[1], own[2]};
        own[0] = prev[0]; own[1] = prev[1]; own[2] = prev[2];
        if (i >= 2) row_stores(i - 2);
        if (i >= 1) epilogue(i - 1);
        PG8_STAMP(3);
        own[0] = cur[0]; own[1] = cur[1]; own[2] = cur[2];
        load_epi(i);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PG8_STAMP(4);
      }
      __builtin_amdgcn_s_barrier();                                                // Bar(i)
      asm volatile("" ::: "memory");
    }
    PG8_STEP_STAMP(3);
    epilogue(ntile - 1);
    if (ntile >= 2) { row_stores(ntile - 2); relay_stores(ntile - 2); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                                  // Bar E1: the last tile's rows are staged
    asm volatile("" ::: "memory");
    PG8_STEP_STAMP(4);
    row_stores(ntile - 1);
    relay_stores(ntile - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              // this step's stores have all been acknowledged
    PG8_STEP_STAMP(5);
    __builtin_amdgcn_s_barrier();                                                  // Bar E2, then lane 0 publishes the step
    asm volatile("" ::: "memory");
    PG8_STEP_STAMP(6);
    if constexpr (HS) {
      if (tid == 0) __hip_atomic_fetch_add(cnt_own, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int H, int KO>
__global__ __launch_bounds__(512) void dec_fwd8_persist_kernel(DecFwdPersist a, Sched sd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int g, l, c;
  if (!persist_map(blockIdx.x, a.NG, Geo8<H>::NCU, g, l, c)) return;
  int Tg = 0;                               // steps of this group = #t with B_t > g
  while (Tg < sd.T && sd.off[Tg + 1] - sd.off[Tg] > g) ++Tg;
  const bool hside = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4;     // wave-uniform, and known to be: a scalar branch
  if (hside) {
    if (l == 0) fwd8_wave<H, 0, 0, KO>(a, sd, g, l, c, smem, Tg);
    else if (l == a.L - 1) fwd8_wave<H, 2, 0, KO>(a, sd, g, l, c, smem, Tg);
    else fwd8_wave<H, 1, 0, KO>(a, sd, g, l, c, smem, Tg);
  } else {
    if (l == 0) fwd8_wave<H, 0, 1, KO>(a, sd, g, l, c, smem, Tg);
    else if (l == a.L - 1) fwd8_wave<H, 2, 1, KO>(a, sd, g, l, c, smem, Tg);
    else fwd8_wave<H, 1, 1, KO>(a, sd, g, l, c, smem, Tg);
  }
}

}  // namespace

// The training configuration of launch_dec_fwd_persist (pg_persist_dec.hip), which has validated the arguments, chosen the
// group count, zeroed the counters and set the give-up words; PG_FWD8_OFF=1 keeps the 4-wave kernel (A/B aid).
bool dec_fwd8_enabled() {
  static const bool off = getenv("PG_FWD8_OFF") && atoi(getenv("PG_FWD8_OFF")) != 0;
  return !off;
}
int launch_dec_fwd8(DecFwdPersist& a, const Sched& sd, int grid, hipStream_t stream) {
  using G = Geo8<512>;
  const void* kern = reinterpret_cast<const void*>(dec_fwd8_persist_kernel<512, 0>);
#ifdef PG_FWD8_KNOCKOUTS
  static const int ko = getenv("PG_FWD8_KO") ? atoi(getenv("PG_FWD8_KO")) : 0;
  switch (ko) {
    case 1: kern = reinterpret_cast<const void*>(dec_fwd8_persist_kernel<512, 1>); break;
    case 2: kern = reinterpret_cast<const void*>(dec_fwd8_persist_kernel<512, 2>); break;
    case 4: kern = reinterpret_cast<const void*>(dec_fwd8_persist_kernel<512, 4>); break;
    case 12: kern = reinterpret_cast<const void*>(dec_fwd8_persist_kernel<512, 12>); break;
    case 15: kern = reinterpret_cast<const void*>(dec_fwd8_persist_kernel<512, 15>); break;
    case 16: kern = reinterpret_cast<const void*>(dec_fwd8_persist_kernel<512, 16>); break;
    case 32: kern = reinterpret_cast<const void*>(dec_fwd8_persist_kernel<512, 32>); break;
    default: break;
  }
#endif
  static const void* attr_done = nullptr;
  if (attr_done != kern) {
    PG_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM));
    attr_done = kern;
  }
  a.sv_lane = 2;   // the pair-split lane order of the saved activations (see the header comment); the caller records it
  void* kargs[] = {&a, const_cast<Sched*>(&sd)};
  PG_CHECK_HIP(hipLaunchKernel(kern, dim3(grid), dim3(512), kargs, (size_t)G::SMEM, stream));
  PG_CHECK_LAUNCH();
  return PG_OK;
}
