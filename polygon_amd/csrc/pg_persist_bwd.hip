// Group-persistent decoder recurrence, backward (BPTT); the mirror image of pg_persist_dec.hip (see pg_persist.h).
//
// Cell (l, t):  dh = z.dh_{t+1} (carry) + dGH_{l,t+1} . W_hh_l  +  mask_l . (dGI_{l+1,t} . W_ih_{l+1})   (top layer: + dout)
// followed by the gate gradients from the saved activations -> dGI_{l,t}, dGH_{l,t}, carry.  Both contractions run over
// the 3H gate columns, so a workgroup (group g, layer l, unit slice c) that owns 64 OUTPUT units keeps the matching rows of
// W_hh_l^T and W_ih_{l+1}^T stationary: per wave 16 units x (48 + 32) K-steps of MFMA fragments in registers (320 VGPRs),
// the last 16 K-steps of W_ih_{l+1}^T (the n-gate columns) of the slice in LDS (65 KB) - the budget of the forward kernel.
//
// Activations are 3x wider than in the forward pass (a row is [dGH | dGI_up] = 6 KB), so a 16-row tile is streamed in six
// 16-KB PARTS (16 rows x 512 gate columns, one LDS-DMA wave instruction per row) through a ring of NSLOT slots; part q+D+1
// is issued when part q starts, every wave waits with a COUNTED vmcnt for its own four DMA instructions of part q, and one
// workgroup barrier per part publishes them (and proves that the slot about to be overwritten is no longer read).
//
// Inter-workgroup protocol: exactly the forward kernel's (write-through stores, vmcnt drain, barrier, one agent-scope
// counter increment per workgroup and step; consumers poll, acquire-fence, and read published rows with sc1 loads only).
// Layer l, step t waits for (l, t+1) and (l+1, t): the layers pipeline top-down by themselves.
#include <hip/hip_runtime.h>
#include "pg_mma.h"
#include "pg_persist.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

namespace {

constexpr int SYNC_STRIDE = 16;
constexpr int SYNC_ERR = PG_MAX_LAYERS * PG_PERSIST_MAX_GROUPS * SYNC_STRIDE;
static_assert(SYNC_ERR + 1 <= PG_PERSIST_SYNC_WORDS, "sync block too small");
constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr int AUX_SC1 = 16;
constexpr int PART_BYTES = 16 * 1024;   // 16 rows x 512 bf16

__device__ __forceinline__ f32x4 mma16(const bf16x8& w, const bf16x8& x, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
}
// the stationary weight fragment straight from an AGPR, hazard nops inside (see pg_persist_dec.hip for the why)
__device__ __forceinline__ f32x4 mma16a(const bf16x8& w, const bf16x8& x, f32x4 c) {
  asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(w), "v"(x));
  return c;
}
// the same without the nop: for an MFMA whose accumulator was last written by an MFMA (every one but a tile's first per
// accumulator) - the 96 nops of a tile were 6 % of its instructions, and one wave per SIMD issues one instruction at a time
__device__ __forceinline__ f32x4 mma16a_nn(const bf16x8& w, const bf16x8& x, f32x4 c) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(w), "v"(x));
  return c;
}
// A group of four K-steps as ONE asm statement: hipcc then waits once for the group's four LDS fragments (left alone it puts
// an exact lgkmcnt wait in front of every single-MFMA asm, and a nop between asm statements that share an accumulator) -
// ~100 fewer instructions per tile.  Accumulation on the same destination two MFMAs apart is interlocked by the hardware.
// The `s_nop 0` between the MFMAs is not a hazard nop: back-to-back 16x16x32 MFMAs of a lone wave issue every 45 cycles, with
// one instruction in between every 33 (tools/mfma_rate.hip).
// FIRST: the tile's first group, whose accumulators were just zeroed by VALU moves (2 wait states).  WA: the weight fragments
// live in AGPRs ("a") or in VGPRs / come from LDS ("v").
#define PG_MMA4_BODY                                               \
  "v_mfma_f32_16x16x32_bf16 %0, %2, %6, %0\n\ts_nop 0\n\t"       \
  "v_mfma_f32_16x16x32_bf16 %1, %3, %7, %1\n\ts_nop 0\n\t"       \
  "v_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\ts_nop 0\n\t"       \
  "v_mfma_f32_16x16x32_bf16 %1, %5, %9, %1\n\ts_nop 0"
template <bool FIRST, bool WA>
__device__ __forceinline__ void mma16_x4(f32x4& c0, f32x4& c1, const bf16x8& w0, const bf16x8& w1, const bf16x8& w2,
                                         const bf16x8& w3, const bf16x8& x0, const bf16x8& x1, const bf16x8& x2, const bf16x8& x3) {
  // FIRST also for every group of the instrumented instantiations (pg_mma.h, mma12_aaa)
  if constexpr (WA) {
    if constexpr (FIRST)
      asm("s_nop 1\n\t" PG_MMA4_BODY : "+v"(c0), "+v"(c1) : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    else
      asm(PG_MMA4_BODY : "+v"(c0), "+v"(c1) : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
  } else {
    if constexpr (FIRST)
      asm("s_nop 1\n\t" PG_MMA4_BODY : "+v"(c0), "+v"(c1) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    else
      asm(PG_MMA4_BODY : "+v"(c0), "+v"(c1) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
  }
}
// matrix-pipe write -> VALU read wait states; tied to the accumulators so that every MFMA asm is ordered before it
#define PG_MFMA_DRAIN(a0, a1, a2, a3) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3))
__device__ __forceinline__ bf16x4 to_bf16x4(const float (&v)[4]) {
  bf16x4 o;
  o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
  return o;
}
template <int AUXS>
__device__ __forceinline__ void store_pub_b64(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const bf16x4& v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)byte_off, 0, AUXS);
}

__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned target, unsigned* err, unsigned* sticky, unsigned limit) {
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

// One part's share of a wave: four 1-KB rows by LDS-DMA, 16 bytes per lane from (base + 32-bit lane offset) to the LDS rows
// m, m + 1024, ...  The instruction offset moves BOTH addresses, so the caller's lane offsets carry -i * 1024.  Issued from
// asm with a scalar base, a row costs 3 instructions (live test, select, DMA) instead of the ~14 of a 64-bit address per row
// - with one wave per SIMD every instruction is a quarter of an MFMA slot - and hipcc, which does not see the transfers,
// adds no waits of its own for them.
template <int AUXL>
__device__ __forceinline__ void dma_rows4(unsigned v0, unsigned v1, unsigned v2, unsigned v3, const char* base, unsigned m) {
  if constexpr (AUXL != 0)
    asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %4 sc1\n\t"
                 "global_load_lds_dwordx4 %1, %4 offset:1024 sc1\n\t"
                 "global_load_lds_dwordx4 %2, %4 offset:2048 sc1\n\t"
                 "global_load_lds_dwordx4 %3, %4 offset:3072 sc1" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(base), "s"(m)
                 : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %4\n\t"
                 "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, %4 offset:2048\n\t"
                 "global_load_lds_dwordx4 %3, %4 offset:3072" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(base), "s"(m)
                 : "memory", "m0");
}

// Epilogue inputs loaded from asm (32-bit lane offset, scalar base): hipcc, which does not see the LDS-DMA transfers, would
// guard a load of its own with vmcnt(0) at the first use and drain the ring once per tile; loaded here and waited for with
// the exact count (epi_wait) the ring stays in flight.  The destination registers must not be touched before epi_wait:
// tools/check_mfma_hazard.py checks the listing for that.
__device__ __forceinline__ void epi_ld16(u32x4& d, unsigned voff, const void* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2 ; PG_EPI_LOAD" : "=v"(d) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void epi_ld8(u32x2& d, unsigned voff, const void* base) {
  asm volatile("global_load_dwordx2 %0, %1, %2 ; PG_EPI_LOAD" : "=v"(d) : "v"(voff), "s"(base) : "memory");
}
// all = every one of the tile's NP parts was issued behind the loads (4 LDS-DMA instructions each): they may stay in
// flight; else the step is ending and everything is waited for
template <int NP>
__device__ __forceinline__ void epi_wait(bool all, u32x4& a0, u32x4& a1, u32x4& a2, u32x4& a3, u32x2& a4) {
  static_assert(NP == 6 || NP == 3, "parts per tile");
  // the waits carry no register operands (hipcc would set up such operands with copies IN FRONT of the wait); the loaded
  // registers are handed to the compiler by the empty asm behind the join
  if (all) {
    if constexpr (NP == 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("; PG_EPI_WAIT" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4)::"memory");
}

// every wave issues 4 LDS-DMA instructions per part; `younger` = parts issued after the one that must have landed.
// Counting fewer outstanding operations than there are (the epilogue's own loads and stores are not counted) only makes
// the wait stricter: vector memory operations of one wave complete in issue order.
__device__ __forceinline__ void wait_parts(int younger) {
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
  }
}

// development aid: workgroup (traced group stamps[2], traced layer, slice 0) records wall-clock stamps (100 MHz) of its steps 8..15.  Stamps and
// the knock-out switches (PG_PERSIST_DBG) live in the DBG instantiation only: as run-time tests they cost the production
// kernel 6-8 scalar branches per part.
#define PG_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if (DBG && a.stamps && tid == 0 && g == (int)a.stamps[2] && c == 0 && l == (int)a.stamps[0] && s >= (int)a.stamps[1] && s < (int)a.stamps[1] + 8) \
      a.stamps[8 + (s - (int)a.stamps[1]) * 8 + (slot)] = wall_clock64();                                               \
  } while (0)

template <int H, bool TOP> struct BwdGeo {
  static constexpr int KS = 3 * H / 32;          // MFMA K steps over the 3H gate columns
  static constexpr int KSP = 16;                 // K steps per part
  static constexpr int KS_REG_UP = 2 * H / 32;   // upper-layer K steps held in registers (r, z columns); the rest in LDS
  static constexpr int NCU = H / 64;
  static constexpr int ROWW = H * 2;             // LDS row of the n-gate columns of W_ih_up^T: unpadded, chunk q of row u at q ^ (u & 15)
  static constexpr int WN_BYTES = TOP ? 0 : 64 * ROWW;
  static constexpr int NSLOT = TOP ? 8 : 4;     // measured: 3 slots 1402 us, 4 slots 1292, 5 slots 1346 (the epilogue drains what is in flight)
  static constexpr int D = NSLOT - 2;            // parts in flight behind the one being consumed
  static constexpr int STG_BYTES = 4 * 16 * 128;  // staging of a tile's gate gradients: dr, dz, dn.r, dn x 16 rows x 64 units (bf16)
  static constexpr int SMEM = WN_BYTES + NSLOT * PART_BYTES + STG_BYTES + 16;
};

template <int V> struct IC { static constexpr int value = V; };

// DBG: stamps and knock-out switches.  FAST: the training configuration (dropout between layers, saved activations in
// lane order) known at compile time.  The step body is compiled three times - first step (no recurrent operand), steady
// state, last step (t = -1, the gradient of h0) - so that the tile loop has no block-uniform branches left: one wave per
// SIMD pays for every scalar branch with idle matrix pipes.
// SVL (FAST only): lane order of the saved activations - 1: (r, z) / (n, hn) pairs as the 4-wave forward kernel writes
// them, 2: the pair-split order of the two-waves-per-SIMD forward kernel (pg_persist_dec8.hip: {r, z, n, hn} of units 0-1 in
// the first region, of units 2-3 in the second).  The other instantiations read a.sv_lane at run time.
template <int H, bool TOP, int AUXL, bool DBG, bool FAST, int SVL>
__device__ __forceinline__ void dec_bwd_body(const DecBwdPersist& a, const Sched& sd, int g, int l, int c, char* smem) {
  using G = BwdGeo<H, TOP>;
  // The knock-out switches (PG_PERSIST_DBG) must never meet the exact-count waits of the FAST path: bit 1 drops the
  // LDS-DMA, and an `s_waitcnt vmcnt(24)` that assumes 24 younger transfers would then let the epilogue's own loads land
  // in registers already reused (a GPU fault).  The DBG instantiations load the epilogue inputs from C++ (hipcc's own
  // conservative waits) and skip wait_parts with the DMA; this assert keeps it that way.
  static_assert(!(DBG && FAST), "the instrumented instantiation must not use the exact-count epilogue waits");
  const int dbg = DBG ? a.dbg : 0;
  constexpr int NCU = G::NCU, ROWW = G::ROWW, NSLOT = G::NSLOT, D = G::D;
  static_assert(H == 512, "one LDS-DMA instruction (64 lanes x 16 B) must cover exactly 512 gate columns of one row");
  static_assert(D <= 6, "wait_parts covers at most 6 parts in flight");
  char* sWn = smem;
  char* sR = smem + G::WN_BYTES;
  char* sStage = smem + G::WN_BYTES + NSLOT * PART_BYTES;
  volatile unsigned* sFlag = reinterpret_cast<volatile unsigned*>(sStage + G::STG_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4, hi = lr >> 2;
  const int NG = a.NG;
  const int uw = c * 64 + wave * 16;
  const int u0 = uw + lq * 4;
  int Tg = 0;
  while (Tg < sd.T && sd.off[Tg + 1] - sd.off[Tg] > g) ++Tg;
  if (Tg == 0) return;   // no sequence of this group (B <= g): nothing to do, nobody waits for this workgroup's counters

  // ---- stationary weights
  const bf16_t* WhhT = a.WhhT[l];
  bf16x8 wh[G::KS];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks)
    wh[ks] = *reinterpret_cast<const bf16x8*>(WhhT + (long)(uw + lr) * 3 * H + (ks * 4 + lq) * 8);
  bf16x8 wu[TOP ? 1 : G::KS_REG_UP];
  if constexpr (!TOP) {
    const bf16_t* WihT = a.WihT[l + 1];
#pragma unroll
    for (int ks = 0; ks < G::KS_REG_UP; ++ks)
      wu[ks] = *reinterpret_cast<const bf16x8*>(WihT + (long)(uw + lr) * 3 * H + (ks * 4 + lq) * 8);
    for (int id = tid; id < 64 * (H / 8); id += 256) {
      const int u = id / (H / 8), ch = id % (H / 8);
      *reinterpret_cast<uint4*>(sWn + u * ROWW + ((ch ^ (u & 15)) << 4)) =
          *reinterpret_cast<const uint4*>(WihT + (long)(c * 64 + u) * 3 * H + 2 * H + ch * 8);
    }
  }
  const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  const bool drop = FAST ? !TOP : (!TOP && a.p_drop > 0.0f);
  const bool sv_lane = FAST ? true : a.sv_lane != 0;
  const bool sv_split = FAST ? SVL == 2 : a.sv_lane == 2;
  bf16_t* dGI = a.dGI[l];
  bf16_t* dGH = a.dGH[l];
  const bf16_t* dGIup = TOP ? nullptr : a.dGI[l + 1];
  float* carry = a.carry[l];
  unsigned* cnt_own = a.sync + (g * PG_MAX_LAYERS + l) * SYNC_STRIDE;
  unsigned* cnt_up = a.sync + (g * PG_MAX_LAYERS + (TOP ? l : l + 1)) * SYNC_STRIDE;
  unsigned* err = a.sync + SYNC_ERR;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  f32x4 sdr = f32x4{0.f, 0.f, 0.f, 0.f}, sdz = sdr, sdnr = sdr, sdn = sdr;   // running sums for the bias gradients
  // LDS-DMA lane offsets of this wave's rows wave*4 + i of a tile (bytes from the operand's column block - 3072): the
  // XOR-swizzled chunk of the lane, + 3072 - i*1024 for dma_rows4's instruction offsets; `vdead` addresses sequence g
  // (finished sequences re-read a valid row; their columns are masked), `vrow` the row's own sequence in the tile at hand
  unsigned vdead[4], vrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) vdead[i] = (unsigned)g * (3 * H * 2) + ((lane ^ (wave * 4 + i)) * 16) + 3072 - i * 1024;
  const unsigned tile_step = 16u * NG * (3 * H * 2);
  const unsigned lds_ring = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)sR) + wave * 4096;
  __syncthreads();

  auto step = [&](int s, auto PHc) __attribute__((always_inline)) -> bool {
    constexpr int PH = decltype(PHc)::value;       // 0: first step, 1: steady state, 2: last step
    const int t = Tg - 1 - s;                      // t = -1: the gradient of h0 (PH == 2)
    constexpr bool has_own = PH != 0;              // step t+1 of this group exists
    constexpr bool has_up = !TOP && PH != 2;
    constexpr bool fin = PH == 2;
    PG_STAMP(0);
    // ---- wait for dGH_{l,t+1} (own layer) and dGI_{l+1,t} (layer above)
    if (wave == 0) {   // (wave-uniform) both counters polled at once: pg_wait_ge2
      const bool ok = pg_wait_ge2(cnt_own, (unsigned)(NCU * s), cnt_up, (unsigned)(NCU * (s + 1)), has_up, err, a.fail, a.spin_limit);
      PG_STAMP(1);
      // No agent acquire (buffer_inv sc1, 0.7-1.2 us per step) here: everything another workgroup publishes is read with
      // sc1 loads (LDS-DMA included), which are served by L2 and never by this CU's vector cache; the other loads of the
      // step read what this workgroup wrote itself or what an earlier launch wrote.  The poll's value has returned
      // (vmcnt(0) below), the barrier orders the other waves' loads behind it, and loads return in issue order.
      if constexpr (AUXL == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // plain payload loads need it
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) *sFlag = ok ? 1u : 0u;
    }
    __syncthreads();
    if (*sFlag == 0u) return false;   // block-uniform: somebody gave up
    PG_STAMP(2);
    const int tt = fin ? 0 : t;
    const int o = sd.off[tt], Bt = sd.off[tt + 1] - o;             // rows of this step (t = -1: those of step 0)
    const int o1 = fin ? 0 : sd.off[t + 1];                         // rows that carry a recurrent gradient: step t+1
    const int Bt1 = !has_own ? 0 : (fin ? Bt : sd.off[t + 2] - o1);
    const int nlive = (Bt - g + NG - 1) / NG;
    const int ntile = (nlive + 15) >> 4;
    constexpr int Pt = (has_own ? 3 : 0) + (has_up ? 3 : 0);       // parts per tile
    const int nparts = ntile * Pt;
    const bf16_t* own_base = dGH + (long)o1 * 3 * H;
    const bf16_t* up_base = TOP ? nullptr : dGIup + (long)o * 3 * H;
    const __amdgpu_buffer_rsrc_t r_dGI = pg_make_rsrc(dGI + (long)o * 3 * H);
    const __amdgpu_buffer_rsrc_t r_dGH = pg_make_rsrc(dGH + (long)o * 3 * H);

    // The step's parts, tile-major, own parts before upper parts; the ring runs AHEAD = D + 1 parts in front of the consumer.
    // A tile's parts are unrolled, so WHICH part is issued behind part p (p + AHEAD: its operand, column block and tile
    // offset) is known at compile time; only the ring slots, the tile of the issue stream and the live-row count are run-time.
    constexpr int AHEAD = D + 1;
    constexpr int RING = NSLOT * PART_BYTES;
    const char* own_b = reinterpret_cast<const char*>(own_base) - 3072;
    const char* up_b = reinterpret_cast<const char*>(up_base) - 3072;
    const int nl_own = Bt1 > g ? (Bt1 - g + NG - 1) / NG - wave * 4 : -wave * 4;   // live rows of the group, from this wave's first row
    const int nl_up = nlive - wave * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) vrow[i] = vdead[i] + (unsigned)(NG * (wave * 4 + i)) * (3 * H * 2);
    int si = 0, sc = 0;                            // ring offsets (bytes) of the next part to issue / to consume
    int jrow = 0;                                  // 16 x the tile of the issue stream (vrow addresses that tile's rows)
    auto issue_part = [&](auto PPc) __attribute__((always_inline)) {
      constexpr int pp = decltype(PPc)::value;     // part of the issue stream's tile
      constexpr bool own = has_own && pp < 3;
      constexpr int sub = own ? pp : pp - (has_own ? 3 : 0);
      if (!(dbg & 1)) {
        const int rem = (own ? nl_own : nl_up) - jrow;   // row i of this wave is live when i < rem
        dma_rows4<AUXL>(0 < rem ? vrow[0] : vdead[0], 1 < rem ? vrow[1] : vdead[1], 2 < rem ? vrow[2] : vdead[2],
                        3 < rem ? vrow[3] : vdead[3], (own ? own_b : up_b) + sub * 1024, lds_ring + si);
      }
      si += PART_BYTES;
      if (si == RING) si = 0;
      if constexpr (pp == Pt - 1) {
        jrow += 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) vrow[i] += tile_step;
      }
    };
    auto prologue = [&](auto Kc, auto&& self) __attribute__((always_inline)) -> void {
      constexpr int k = decltype(Kc)::value;
      if constexpr (Pt > 0 && k < AHEAD) {
        if (k / Pt < ntile) {                      // block-uniform
          issue_part(IC<k % (Pt > 0 ? Pt : 1)>{});
          self(IC<k + 1>{}, self);
        }
      }
    };
    prologue(IC<0>{}, prologue);
    PG_STAMP(3);
    // one part: wait for its DMAs, publish them, refill the slot freed by the previous part, 16 K-steps
    auto part_begin = [&](int j, auto Pc) __attribute__((always_inline)) -> const char* {
      constexpr int pt = decltype(Pc)::value;      // part of tile j
      constexpr int c = Pt - 1 - pt;               // parts of this tile behind it
      if (!(dbg & 1)) {
        if constexpr (c >= D) {
          wait_parts(D);
        } else {
          const int younger = (ntile - 1 - j) * Pt + c;
          if (younger >= D) wait_parts(D);
          else wait_parts(younger);
        }
      }
      // a bare s_barrier: __syncthreads() carries a workgroup release fence = vmcnt(0), which would drain the very DMAs
      // this pipeline keeps in flight.  Nothing but the DMA'd slot is communicated here (the wave's own LDS reads of the
      // previous part were consumed by its MFMAs), so the counted wait above is the only ordering needed.
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      constexpr int k = pt + AHEAD, dj = k / (Pt > 0 ? Pt : 1), pp = k % (Pt > 0 ? Pt : 1);
      if (dj == 0 || j + dj < ntile) issue_part(IC<pp>{});   // block-uniform
      const char* A = sR + sc + lr * 1024 + ((lq ^ lr) & 3) * 16;
      sc += PART_BYTES;
      if (sc == RING) sc = 0;
      return A;
    };

    for (int j = 0; j < ntile; ++j) {
      const int b = g + NG * (16 * j + lr);
      const bool live = b < Bt, live1 = b < Bt1;
      const int bs = live ? b : g;
      // epilogue inputs of this lane's sequence, fetched under the tile's MFMAs
      f32x4 cr = f32x4{0.f, 0.f, 0.f, 0.f}, dtop = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x4 svr, svz, svn, svhn, svhp;
      constexpr bool ASMLD = FAST && !fin && Pt > 0;   // (FAST: lane-order saves, no knock-out switches)
      u32x4 q_cr = u32x4{0u, 0u, 0u, 0u}, q_top = q_cr, q_rz = q_cr, q_nh = q_cr;
      u32x2 q_hp = u32x2{0u, 0u};
      if constexpr (ASMLD) {
        const unsigned sidx = (unsigned)((o + bs) * H + u0);     // < 2^31 / 4: N * H elements
        if constexpr (has_own) epi_ld16(q_cr, (unsigned)(bs * H + u0) * 4u, carry);
        if constexpr (TOP) epi_ld16(q_top, sidx * 4u, a.dout);
        const unsigned sl = (unsigned)(((sv_tile_slot(o, t, NG, j, g) * NCU + c) * 256 + wave * 64 + lane) * 16);
        epi_ld16(q_rz, sl, a.sv[l][0]);
        epi_ld16(q_nh, sl, a.sv[l][2]);
        epi_ld8(q_hp, sidx * 2u, a.sv[l][4]);
      } else {
        if (has_own && !(dbg & 4)) cr = *reinterpret_cast<const f32x4*>(carry + (long)bs * H + u0);
        if (!fin && !(dbg & 4)) {
          const long sidx = (long)(o + bs) * H + u0;
          if constexpr (TOP) dtop = *reinterpret_cast<const f32x4*>(a.dout + sidx);
          // lane order, as the forward kernel wrote them (pg_persist.h, sv_tile_slot)
          if (sv_lane) {   // kernel-uniform: (r, z) and (n, hn) pairs, 16 bytes per lane each (pg_persist_dec.hip)
            const long sl = ((sv_tile_slot(o, t, NG, j, g) * NCU + c) * 256 + wave * 64 + lane) * 8;
            const u32x4 prz = *reinterpret_cast<const u32x4*>(a.sv[l][0] + sl);
            const u32x4 pnh = *reinterpret_cast<const u32x4*>(a.sv[l][2] + sl);
            if (sv_split) {   // kernel-uniform
              svr = __builtin_bit_cast(bf16x4, u32x2{prz[0], pnh[0]});
              svz = __builtin_bit_cast(bf16x4, u32x2{prz[1], pnh[1]});
              svn = __builtin_bit_cast(bf16x4, u32x2{prz[2], pnh[2]});
              svhn = __builtin_bit_cast(bf16x4, u32x2{prz[3], pnh[3]});
            } else {
              svr = __builtin_bit_cast(bf16x4, u32x2{prz[0], prz[1]});
              svz = __builtin_bit_cast(bf16x4, u32x2{prz[2], prz[3]});
              svn = __builtin_bit_cast(bf16x4, u32x2{pnh[0], pnh[1]});
              svhn = __builtin_bit_cast(bf16x4, u32x2{pnh[2], pnh[3]});
            }
          } else {
            svr = *reinterpret_cast<const bf16x4*>(a.sv[l][0] + sidx);
            svz = *reinterpret_cast<const bf16x4*>(a.sv[l][1] + sidx);
            svn = *reinterpret_cast<const bf16x4*>(a.sv[l][2] + sidx);
            svhn = *reinterpret_cast<const bf16x4*>(a.sv[l][3] + sidx);
          }
          svhp = *reinterpret_cast<const bf16x4*>(a.sv[l][4] + sidx);
        }
      }
      f32x4 ah[2], au[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { ah[i] = f32x4{0.f, 0.f, 0.f, 0.f}; au[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      if constexpr (has_own) {
        auto own_part = [&](auto Pc) __attribute__((always_inline)) {
          constexpr int p = decltype(Pc)::value;
          const char* A = part_begin(j, Pc);
          if (dbg & 2) return;
          // the part's 16 activation fragments are read in groups of 4, two groups (8 reads, <= 0.5 us of MFMAs) ahead of
          // the MFMAs that consume them; the scheduling barriers keep hipcc from sinking the reads back to their uses
          bf16x8 f[4][4];
#pragma unroll
          for (int gk = 0; gk < 2; ++gk)
#pragma unroll
            for (int i = 0; i < 4; ++i) f[gk][i] = *reinterpret_cast<const bf16x8*>(A + (((gk * 4 + i) ^ hi) << 6));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int gk = 0; gk < 4; ++gk) {
            if (gk + 2 < 4) {
#pragma unroll
              for (int i = 0; i < 4; ++i) f[gk + 2][i] = *reinterpret_cast<const bf16x8*>(A + ((((gk + 2) * 4 + i) ^ hi) << 6));
            }
            __builtin_amdgcn_sched_barrier(0);
            const int k0 = p * 16 + gk * 4;
            if (p == 0 && gk == 0) mma16_x4<true, true>(ah[0], ah[1], wh[k0], wh[k0 + 1], wh[k0 + 2], wh[k0 + 3], f[gk][0], f[gk][1], f[gk][2], f[gk][3]);
            else mma16_x4<DBG, true>(ah[0], ah[1], wh[k0], wh[k0 + 1], wh[k0 + 2], wh[k0 + 3], f[gk][0], f[gk][1], f[gk][2], f[gk][3]);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        own_part(IC<0>{});
        own_part(IC<1>{});
        own_part(IC<2>{});
      }
      if constexpr (!TOP) {
        if constexpr (has_up) {
          constexpr int P0 = has_own ? 3 : 0;      // the upper parts follow the own parts of the tile
          auto up_part = [&](auto Pc) __attribute__((always_inline)) {
            constexpr int p = decltype(Pc)::value;
            const char* A = part_begin(j, IC<P0 + p>{});
            if (dbg & 2) return;
            bf16x8 f[4][4];
#pragma unroll
            for (int gk = 0; gk < 2; ++gk)
#pragma unroll
              for (int i = 0; i < 4; ++i) f[gk][i] = *reinterpret_cast<const bf16x8*>(A + (((gk * 4 + i) ^ hi) << 6));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gk = 0; gk < 4; ++gk) {
              if (gk + 2 < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) f[gk + 2][i] = *reinterpret_cast<const bf16x8*>(A + ((((gk + 2) * 4 + i) ^ hi) << 6));
              }
              __builtin_amdgcn_sched_barrier(0);
              const int k0 = p * 16 + gk * 4;
              if (p == 0 && gk == 0) mma16_x4<true, true>(au[0], au[1], wu[k0], wu[k0 + 1], wu[k0 + 2], wu[k0 + 3], f[gk][0], f[gk][1], f[gk][2], f[gk][3]);
              else if (p == 0) mma16_x4<DBG, true>(au[0], au[1], wu[k0], wu[k0 + 1], wu[k0 + 2], wu[k0 + 3], f[gk][0], f[gk][1], f[gk][2], f[gk][3]);
              else mma16_x4<DBG, false>(au[0], au[1], wu[k0], wu[k0 + 1], wu[k0 + 2], wu[k0 + 3], f[gk][0], f[gk][1], f[gk][2], f[gk][3]);
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          up_part(IC<0>{});
          up_part(IC<1>{});
          {
            const char* A = part_begin(j, IC<P0 + 2>{});
            const char* Wn = sWn + (wave * 16 + lr) * ROWW + ((lq ^ lr) & 3) * 16;   // chunk ks*4+lq of row lr sits at (ks*4+lq) ^ lr
            if (!(dbg & 2)) {
            // activation and weight fragments both come from LDS here: groups of 4 K-steps, one group ahead
            bf16x8 xa[4], wa[4], xb[4], wb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              xa[i] = *reinterpret_cast<const bf16x8*>(A + ((i ^ hi) << 6));
              wa[i] = *reinterpret_cast<const bf16x8*>(Wn + ((i ^ hi) << 6));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gk = 0; gk < 4; ++gk) {
              if (gk + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const int ks = (gk + 1) * 4 + i;
                  if (gk & 1) {
                    xa[i] = *reinterpret_cast<const bf16x8*>(A + ((ks ^ hi) << 6));
                    wa[i] = *reinterpret_cast<const bf16x8*>(Wn + ((ks ^ hi) << 6));
                  } else {
                    xb[i] = *reinterpret_cast<const bf16x8*>(A + ((ks ^ hi) << 6));
                    wb[i] = *reinterpret_cast<const bf16x8*>(Wn + ((ks ^ hi) << 6));
                  }
                }
              }
              __builtin_amdgcn_sched_barrier(0);
              if (gk & 1) mma16_x4<DBG, false>(au[0], au[1], wb[0], wb[1], wb[2], wb[3], xb[0], xb[1], xb[2], xb[3]);
              else mma16_x4<DBG, false>(au[0], au[1], wa[0], wa[1], wa[2], wa[3], xa[0], xa[1], xa[2], xa[3]);
              __builtin_amdgcn_sched_barrier(0);
            }
            }
          }
        }
      }
      PG_MFMA_DRAIN(ah[0], ah[1], au[0], au[1]);
      if (j == 0) PG_STAMP(4);
      if constexpr (ASMLD) {
        epi_wait<Pt>(j * Pt + Pt + AHEAD <= nparts, q_cr, q_top, q_rz, q_nh, q_hp);
        cr = __builtin_bit_cast(f32x4, q_cr);
        dtop = __builtin_bit_cast(f32x4, q_top);
        if constexpr (SVL == 2) {
          svr = __builtin_bit_cast(bf16x4, u32x2{q_rz[0], q_nh[0]});
          svz = __builtin_bit_cast(bf16x4, u32x2{q_rz[1], q_nh[1]});
          svn = __builtin_bit_cast(bf16x4, u32x2{q_rz[2], q_nh[2]});
          svhn = __builtin_bit_cast(bf16x4, u32x2{q_rz[3], q_nh[3]});
        } else {
          svr = __builtin_bit_cast(bf16x4, u32x2{q_rz[0], q_rz[1]});
          svz = __builtin_bit_cast(bf16x4, u32x2{q_rz[2], q_rz[3]});
          svn = __builtin_bit_cast(bf16x4, u32x2{q_nh[0], q_nh[1]});
          svhn = __builtin_bit_cast(bf16x4, u32x2{q_nh[2], q_nh[3]});
        }
        svhp = __builtin_bit_cast(bf16x4, q_hp);
      }
      // ---- epilogue on 4 units of sequence b
      float dh[4];
      float m[4] = {1.0f, 1.0f, 1.0f, 1.0f};
      if (drop && !fin) dropout_scale4(a.seed, l, t, b * a.b_mul + a.b_add, u0 >> 2, a.p_drop, inv_keep, m);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d = live1 ? cr[e] + ah[0][e] + ah[1][e] : 0.0f;
        if constexpr (TOP) d += dtop[e];
        else d += m[e] * (au[0][e] + au[1][e]);
        dh[e] = d;
      }
      if constexpr (fin) {
        if (live) *reinterpret_cast<f32x4*>(a.dh0[l] + (long)b * H + u0) = f32x4{dh[0], dh[1], dh[2], dh[3]};
        continue;
      }
      if (live && !((dbg & 4) && dh[0] != 12345.678f)) {
        float dr[4], dz[4], dn[4], dnr[4], dhz[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r = (float)svr[e], z = (float)svz[e], n = (float)svn[e], hn = (float)svhn[e], hp = (float)svhp[e];
          dn[e] = dh[e] * (1.0f - z) * (1.0f - n * n);
          dz[e] = dh[e] * (hp - n) * z * (1.0f - z);
          dr[e] = dn[e] * hn * r * (1.0f - r);
          dnr[e] = dn[e] * r;
          dhz[e] = dh[e] * z;
          sdr[e] += dr[e]; sdz[e] += dz[e]; sdnr[e] += dnr[e]; sdn[e] += dn[e];
        }
        *reinterpret_cast<f32x4*>(carry + (long)b * H + u0) = f32x4{dhz[0], dhz[1], dhz[2], dhz[3]};
        // The published gate gradients leave through LDS: a lane holds 4 units of one row, so direct stores are 16 rows
        // x 32 bytes per wave instruction (16 cache-line accesses, ~35 CU cycles each, 6 of them per tile); staged, they
        // go out as 128-byte row segments.  Tile [gate][row][64 units], 16-byte chunk q of row r at position q ^ (r & 7).
        char* S = sStage + lr * 128 + (((wave * 2 + (lq >> 1)) ^ (lr & 7)) << 4) + (lq & 1) * 8;
        *reinterpret_cast<bf16x4*>(S) = to_bf16x4(dr);
        *reinterpret_cast<bf16x4*>(S + 2048) = to_bf16x4(dz);
        *reinterpret_cast<bf16x4*>(S + 4096) = to_bf16x4(dnr);
        *reinterpret_cast<bf16x4*>(S + 6144) = to_bf16x4(dn);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!(dbg & 4)) {
        // 2 arrays x 3 gates x 16 rows x 8 segments = 768 segment stores, 3 per thread; array and gate are wave-uniform
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int wq = k * 4 + wave;                       // 0..11
          const int arr = wq / 6, gate = (wq % 6) >> 1;
          const int row = ((wq & 1) << 3) + (lane >> 3), seg = lane & 7;
          const int tile = gate < 2 ? gate : (arr ? 3 : 2);
          const u32x4 v = *reinterpret_cast<const u32x4*>(sStage + tile * 2048 + row * 128 + ((seg ^ (row & 7)) << 4));
          const int brow = g + NG * (16 * j + row);
          const unsigned off = brow < Bt ? (unsigned)((((long)brow * 3 * H + gate * H + c * 64) * 2) + seg * 16) : 0x80000000u;
          if (arr == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r_dGH, (int)off, 0, AUXL);
          else __builtin_amdgcn_raw_buffer_store_b128(v, r_dGI, (int)off, 0, AUXL);
        }
      }
      // the staging tile is rewritten by the next tile's epilogue, behind that tile's part barriers; a tile without
      // parts (top layer, first step) needs one of its own
      if (Pt == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    PG_STAMP(5);
    // ---- publish step t of this unit slice (also: every wave is done with the ring before the next step refills it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PG_STAMP(6);
    if (tid == 0 && !fin) {
      if constexpr (AUXL == 0) {   // plain payload stores: lane-0 agent release (L2 write-back) before the arrival
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(cnt_own, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PG_STAMP(7);
    return true;
  };
  if (!step(0, IC<0>{})) return;
  for (int s = 1; s < Tg; ++s)
    if (!step(s, IC<1>{})) return;
  if (!step(Tg, IC<2>{})) return;
  // ---- bias gradients: the 16 lanes that share this lane's units (one per sequence row of the tile) are summed, then one
  // lane adds 4 units x 3 gates per array
  if (a.dbhh[l] || a.dbih[l]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {
        sdr[e] += __shfl_xor(sdr[e], o, 16); sdz[e] += __shfl_xor(sdz[e], o, 16);
        sdnr[e] += __shfl_xor(sdnr[e], o, 16); sdn[e] += __shfl_xor(sdn[e], o, 16);
      }
    if (lr == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (a.dbhh[l]) {
          atomicAdd(a.dbhh[l] + u0 + e, sdr[e]); atomicAdd(a.dbhh[l] + H + u0 + e, sdz[e]); atomicAdd(a.dbhh[l] + 2 * H + u0 + e, sdnr[e]);
        }
        if (a.dbih[l]) {
          atomicAdd(a.dbih[l] + u0 + e, sdr[e]); atomicAdd(a.dbih[l] + H + u0 + e, sdz[e]); atomicAdd(a.dbih[l] + 2 * H + u0 + e, sdn[e]);
        }
      }
    }
  }
}

template <int H, int AUXL, bool DBG, bool FAST, int SVL = 1>
__global__ __launch_bounds__(256) void dec_bwd_persist_kernel(DecBwdPersist a, Sched sd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NCU = H / 64;
  int g, li, c;   // blockIdx % 8 labels the XCD: a group's workgroups (or at least a layer's) share one
  if (!persist_map(blockIdx.x, a.NG, NCU, g, li, c)) return;
  const int l = a.L - 1 - li;                               // top layer first: it is the head of the dependency chain
  if (l == a.L - 1) dec_bwd_body<H, true, AUXL, DBG, FAST, SVL>(a, sd, g, l, c, smem);
  else dec_bwd_body<H, false, AUXL, DBG, FAST, SVL>(a, sd, g, l, c, smem);
}

// dh0 = sum over layers of the per-layer h0 gradients (every layer starts from the same decoder_lat(z))
__global__ void dh0_sum_kernel(DecBwdPersist a, long n, float* out32, bf16_t* out_op) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(a.dh0[0] + i);
  for (int l = 1; l < a.L; ++l) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.dh0[l] + i);
    s += v;
  }
  *reinterpret_cast<f32x4*>(out32 + i) = s;
  const float v[4] = {s[0], s[1], s[2], s[3]};
  *reinterpret_cast<bf16x4*>(out_op + i) = to_bf16x4(v);
}

}  // namespace

int launch_dec_bwd_persist(DecBwdPersist& a, const Sched& sd, int H, float* dh0_32, bf16_t* dh0_op, hipStream_t stream,
                           int host_flags) {
  PG_REQUIRE(dec_persist_supported(H, a.L, a.B), "dec_bwd_persist: shape H=%d L=%d not covered", H, a.L);
  PG_REQUIRE(a.sync && a.dout && dh0_32 && dh0_op, "dec_bwd_persist: NULL buffer");
  for (int l = 0; l < a.L; ++l)
    PG_REQUIRE(a.WhhT[l] && (l == 0 || a.WihT[l]) && a.dGI[l] && a.dGH[l] && a.carry[l] && a.dh0[l] && a.sv[l][0],
               "dec_bwd_persist: NULL buffer (layer %d)", l);
  using GL = BwdGeo<512, false>;
  using GT = BwdGeo<512, true>;
  constexpr int SMEM = GL::SMEM > GT::SMEM ? GL::SMEM : GT::SMEM;
  a.NG = persist_groups(a.L, a.B);
  PG_TRY(persist_fail_words(&a.fail));
  a.spin_limit = persist_spin_limit();
  a.stamps = persist_stamps_for(1);
  a.dbg = getenv("PG_PERSIST_DBG") ? atoi(getenv("PG_PERSIST_DBG")) : 0;
  if (!(host_flags & PG_PERSIST_SYNC_ZEROED))
    PG_CHECK_HIP(hipMemsetAsync(a.sync, 0, PG_PERSIST_SYNC_WORDS * sizeof(unsigned), stream));
  const bool dbgk = a.dbg != 0 || a.stamps != nullptr;
  const bool plain = persist_dma_plain();
  const bool fast = !dbgk && !plain && a.L >= 2 && a.sv_lane && a.p_drop > 0.0f && !getenv("PG_PERSIST_NO_FAST");
  const void* kern = fast    ? (a.sv_lane == 2 ? reinterpret_cast<const void*>(dec_bwd_persist_kernel<512, AUX_SC1, false, true, 2>)
                                                : reinterpret_cast<const void*>(dec_bwd_persist_kernel<512, AUX_SC1, false, true, 1>))
                     : plain ? (dbgk ? reinterpret_cast<const void*>(dec_bwd_persist_kernel<512, 0, true, false>)
                                     : reinterpret_cast<const void*>(dec_bwd_persist_kernel<512, 0, false, false>))
                             : (dbgk ? reinterpret_cast<const void*>(dec_bwd_persist_kernel<512, AUX_SC1, true, false>)
                                     : reinterpret_cast<const void*>(dec_bwd_persist_kernel<512, AUX_SC1, false, false>));
  static bool attr_done[6] = {false, false, false, false, false, false};
  const int ki = fast ? (a.sv_lane == 2 ? 5 : 4) : plain * 2 + dbgk;
  if (!attr_done[ki]) {
    PG_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_done[ki] = true;
  }
  const int grid = a.NG == 8 ? 8 * a.L * GL::NCU : 256;   // co-residency: one workgroup per CU (checked against the CU count by the forward launcher's rule)
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    PG_CHECK_HIP(hipGetDevice(&dev));
    PG_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  PG_REQUIRE(grid <= n_cu, "dec_bwd_persist: %d workgroups cannot be co-resident on %d CUs", grid, n_cu);
  void* kargs[] = {&a, const_cast<Sched*>(&sd)};
  PG_CHECK_HIP(hipLaunchKernel(kern, dim3(grid), dim3(256), kargs, (size_t)SMEM, stream));
  PG_CHECK_LAUNCH();
  if (!(host_flags & PG_PERSIST_NO_MIRROR)) PG_TRY(persist_fail_mirror(stream));
  const long n = (long)a.B * H;
  hipLaunchKernelGGL(dh0_sum_kernel, dim3(cdiv(n, 4 * 256)), dim3(256), 0, stream, a, n, dh0_32, dh0_op);
  PG_CHECK_LAUNCH();
  return PG_OK;
}
