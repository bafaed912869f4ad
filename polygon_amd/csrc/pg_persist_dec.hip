// Group-persistent decoder recurrence, forward (see pg_persist.h).  bf16 operands, fp32 state and accumulation.
//
// Workgroup = (group g, layer l, unit slice c): 4 waves, wave w owns hidden units c*64 + w*16 .. +15 with all three gates.
// Stationary weights per wave: W_hh rows of its units for r, z, n and W_ih rows for r, z as MFMA fragments in registers
// (5 x 16 K-steps x 4 VGPRs = 320 VGPRs/AGPRs of the 512 a 1-wave-per-SIMD kernel owns); the W_in rows of the workgroup's
// 64 units sit in LDS (65 KB).  Per step the workgroup streams its sequences in tiles of 16 rows x [h_{t-1} | x] (2 KB per
// row) through a double-buffered LDS tile; v_mfma_f32_16x16x32_bf16 takes the weights as first operand, so a lane ends up
// with 4 consecutive units of one sequence and the gate epilogue runs straight on the accumulators.
//
// Inter-workgroup protocol (placement independent; every wait is bounded):
//   producer: h / dropout(h) slices are stored WRITE-THROUGH (sc1), every wave drains vmcnt, workgroup barrier, one lane
//             adds 1 to the (group, layer) counter with an agent-scope atomic;
//   consumer: one lane polls the counters it depends on (relaxed agent-scope loads + s_sleep), then an agent-scope
//             acquire fence, workgroup barrier, and the published rows are read with sc1 loads only.
// Every published row is written exactly once per launch and never read before its counter says so.
#include <hip/hip_runtime.h>
#include "pg_mma.h"
#include "pg_persist.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

namespace {

constexpr int SYNC_STRIDE = 16;                                   // words between counters (one 64-byte line each)
constexpr int SYNC_ERR = PG_MAX_LAYERS * PG_PERSIST_MAX_GROUPS * SYNC_STRIDE;         // give-up word, after the [8][PG_MAX_LAYERS] counters
static_assert(SYNC_ERR + 1 <= PG_PERSIST_SYNC_WORDS, "sync block too small");
constexpr unsigned SPIN_LIMIT = 1u << 21;                         // ~2-4 s of polling before a workgroup gives up
constexpr int AUX_SC1 = 16;

__device__ __forceinline__ f32x4 mma16(const bf16x8& w, const bf16x8& x, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
}
// The same MFMA with its stationary weight operand taken straight from an accumulation register (AGPR).  A wave of these
// kernels holds 320 registers of weight fragments, more than the 256 architectural VGPRs: left to itself hipcc parks
// them in AGPRs and copies each fragment back with four v_accvgpr_read (plus a hazard nop) in front of every MFMA.
// The instruction accepts an AGPR for srcA, so the "a" constraint removes ~200 VALU instructions per tile.  Inline asm is
// opaque to hipcc's hazard recognizer: the caller puts PG_MFMA_DRAIN between the last MFMA and the first VALU read of an
// accumulator (matrix-pipe write -> VALU read needs up to 11 wait states for this 4-pass instruction); LDS-read -> MFMA
// operand dependencies are still tracked (s_waitcnt is inserted for asm operands), and accumulate chains on the same
// destination are interlocked by the hardware (and separated by two other MFMAs here).
__device__ __forceinline__ f32x4 mma16a(const bf16x8& w, const bf16x8& x, f32x4 c) {
  // s_nop 1: where hipcc had to move a fragment into the AGPR (or just zeroed the accumulator) with VALU instructions,
  // the 2 wait states a VALU write needs before an MFMA reads it; hipcc cannot know that this asm is an MFMA
  asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(w), "v"(x));
  return c;
}
// tied to the accumulators: every MFMA that produces them is ordered before it and every consumer after it (a bare
// asm volatile would not order the non-volatile MFMA asms)
#define PG_MFMA_DRAIN(a0, a1, a2, a3) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3))
__device__ __forceinline__ bf16x4 to_bf16x4(const float (&v)[4]) {
  bf16x4 o;
  o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
  return o;
}
// 8-byte write-through store (published data)
template <int AUXS>
__device__ __forceinline__ void store_pub_b64(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const bf16x4& v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)byte_off, 0, AUXS);
}
__device__ __forceinline__ uint4 load_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX_SC1);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// one lane: wait until *p >= target.  false: this or another workgroup gave up (err word set).
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

// development aid: workgroup (g 0, layer STAMP_L, slice 0) records wall-clock stamps (100 MHz) of steps 8..15, 8 slots each
#define PG_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if (DBG && a.stamps && tid == 0 && g == (int)a.stamps[2] && c == 0 && l == (int)a.stamps[0] && t >= (int)a.stamps[1] && t < (int)a.stamps[1] + 8) \
      a.stamps[8 + (t - (int)a.stamps[1]) * 8 + (slot)] = wall_clock64();                                               \
  } while (0)

template <int H> struct DecGeo {
  static constexpr int KS = H / 32;             // MFMA K steps over a K = H contraction
  static constexpr int NCU = H / 64;            // workgroups (unit slices) per layer
  static constexpr int ROWW = H * 2;            // W_in row in LDS: unpadded, 16-byte chunk q of row u at position q ^ (u & 15)
  static constexpr int WIN_BYTES = 64 * ROWW;
  static constexpr int ACT_BYTES = 16 * 2 * H * 2;    // activation tile rows [h_{t-1} | x], unpadded (swizzled); layer 0 uses half
  static constexpr int CARRY_BYTES = 16 * 64 * 4;     // fp32 state slice of the tile's rows (this workgroup's 64 units)
  static constexpr int GATE_BYTES = 3 * CARRY_BYTES;  // layer 0: gather-table rows / z-projection rows, 3 gates
  // layers >= 1: W_in rows + 2 x (activations + carry); layer 0 (no W_in, half the activations) fits its two extra
  // epilogue inputs in the same footprint
  static constexpr int TILE_BYTES = ACT_BYTES + CARRY_BYTES;
  static constexpr int TILE0_BYTES = ACT_BYTES / 2 + CARRY_BYTES + 2 * GATE_BYTES;
  static constexpr int TOK_BYTES = 128 * 4;           // layer 0: the step's tokens of this group (<= 128 sequences)
  // output staging: the bf16 rows a tile publishes (h, dropout(h)) or saves row-major (h_prev), [2 buffers][3 arrays][16 rows]
  // of 64 units, rows padded to 144 bytes; written per lane, read back as 128-byte row segments for coalesced stores
  static constexpr int STG_ROW = 144, STG_ARR = 16 * STG_ROW, STG_BUF = 3 * STG_ARR;
  static constexpr int STG_OFF = WIN_BYTES + 2 * TILE_BYTES;
  static constexpr int SMEM = STG_OFF + 2 * STG_BUF + 16;        // + the broadcast word of the poll result
  static_assert(2 * TILE0_BYTES + TOK_BYTES <= STG_OFF, "layer-0 layout must fit in front of the staging block");
};

// LK: 0 = layer 0, 1 = a middle layer, 2 = the top layer.  DBG: stamps and the PG_PERSIST_DBG knock-out switches (as
// run-time tests they cost the production kernel scalar branches in every tile).  FAST: the training configuration -
// dropout between layers, activations saved in lane order - known at compile time, so that the tile loop has no
// kernel-uniform branches and its store count is a constant.
template <int H, int LK, int AUXL, bool DBG, bool FAST>
__device__ __forceinline__ void dec_fwd_body(const DecFwdPersist& a, const Sched& sd, int g, int l, int c, char* smem) {
  using G = DecGeo<H>;
  constexpr bool L0 = LK == 0;
  const int dbg = DBG ? a.dbg : 0;
  constexpr int KS = G::KS, NCU = G::NCU, ROWW = G::ROWW;
  static_assert(H == 512, "one LDS-DMA instruction (64 lanes x 16 B) must cover exactly one row half");
  char* sWin = smem;
  char* sA = smem + G::WIN_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int NG = a.NG;
  const int uw = c * 64 + wave * 16;        // first unit of this wave
  const int u0 = uw + lq * 4;               // first of this lane's 4 units
  int Tg = 0;                               // steps of this group = #t with B_t > g
  while (Tg < sd.T && sd.off[Tg + 1] - sd.off[Tg] > g) ++Tg;

  // ---- stationary weights
  const bf16_t* Whh = a.Whh[l];
  const bf16_t* Wih = a.Wih[l];
  bf16x8 wh[3][KS];
#pragma unroll
  for (int gt = 0; gt < 3; ++gt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wh[gt][ks] = *reinterpret_cast<const bf16x8*>(Whh + (long)(gt * H + uw + lr) * H + (ks * 4 + lq) * 8);
  bf16x8 wi[2][L0 ? 1 : KS];
  if constexpr (!L0) {
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wi[gt][ks] = *reinterpret_cast<const bf16x8*>(Wih + (long)(gt * H + uw + lr) * H + (ks * 4 + lq) * 8);
    for (int id = tid; id < 64 * (H / 8); id += 256) {
      const int u = id / (H / 8), ch = id % (H / 8);
      *reinterpret_cast<uint4*>(sWin + u * ROWW + ((ch ^ (u & 15)) << 4)) =
          *reinterpret_cast<const uint4*>(Wih + (long)(2 * H + c * 64 + u) * H + ch * 8);
    }
  }
  float bh[4], bgi[3][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bh[e] = a.bhn[l][u0 + e];
#pragma unroll
    for (int gt = 0; gt < 3; ++gt) bgi[gt][e] = L0 ? 0.0f : a.bias_gi[l][gt * H + u0 + e];
  }
  const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  bf16_t* hseq = a.hseq[l];
  bf16_t* hdrop = a.hdrop[l];
  const bf16_t* xsrc = L0 ? nullptr : (a.hdrop[l - 1] ? a.hdrop[l - 1] : a.hseq[l - 1]);
  float* carry = a.carry[l];
  const __amdgpu_buffer_rsrc_t r_hseq = pg_make_rsrc(hseq);
  const __amdgpu_buffer_rsrc_t r_hdrop = pg_make_rsrc(hdrop ? hdrop : hseq);
  unsigned* cnt_own = a.sync + (g * PG_MAX_LAYERS + l) * SYNC_STRIDE;
  unsigned* cnt_low = a.sync + (g * PG_MAX_LAYERS + (L0 ? 0 : l - 1)) * SYNC_STRIDE;
  unsigned* err = a.sync + SYNC_ERR;
  __syncthreads();

  // LDS-DMA staging of one 16-row tile: one wave instruction moves one 1-KB row half ([h_{t-1}] or [x]) straight into
  // LDS (no staging registers).  The destination is lane-linear, so the bank-conflict-free image is made on the SOURCE
  // side: lane p of row r fetches 16-byte chunk p ^ r, i.e. LDS position p of the row holds chunk p ^ (r & 15), and a
  // fragment read of chunk q of row r goes to position q ^ r (16 rows x one 4-chunk group = 16 distinct bank groups).
  // The epilogue's inputs ride along: the fp32 state slice of the tile (16 rows x 256 B) and, in layer 0, the gather-table
  // and z-projection rows (3 gates x 256 B per row each) - a quarter of the rows per wave instruction, the same XOR image.
  constexpr int PARTS = L0 ? 1 : 2;                 // row halves per row
  constexpr int RP_WAVE = 16 * PARTS / 4;           // (row, part) pairs per wave per tile
  constexpr int TILEB = L0 ? G::TILE0_BYTES : G::TILE_BYTES;
  constexpr int ACTB = G::ACT_BYTES / (L0 ? 2 : 1);
  char* sT = L0 ? smem : sA;                        // tile buffers (layer 0 has no W_in block in front)
  int* sTok = reinterpret_cast<int*>(smem + 2 * G::TILE0_BYTES);   // layer 0 only
  volatile unsigned* sFlagT = reinterpret_cast<volatile unsigned*>(smem + G::SMEM - 16);
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const bool has_drop = FAST ? LK != 2 : hdrop != nullptr;                      // kernel-uniform
  const bool has_sv = FAST ? true : a.sv[l][0] != nullptr;
  const bool sv_lane = FAST ? true : a.sv_lane != 0;
  const int narr = 1 + (has_drop ? 1 : 0) + (has_sv ? 1 : 0);                   // row-major bf16 outputs: h, dropout(h), h_prev
  const int NS = 1 + (narr > 2 ? 2 : 1) + (has_sv ? (sv_lane ? 2 : 4) : 0);     // store instructions per wave and tile
  char* sStage = smem + G::STG_OFF;
  const __amdgpu_buffer_rsrc_t r_carry = pg_make_rsrc(carry);
  // vector memory operations of a wave complete in issue order and loads and stores share ONE counter, so a wait is
  // written as "all but the youngest n": n = the stores issued after the DMA that must have landed
  auto wait_all_but = [&](int n) __attribute__((always_inline)) {
    switch (n) {
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };

  for (int t = 0; t < Tg; ++t) {
    PG_STAMP(0);
    // ---- wait for h_{l,t-1} (own layer) and x = h_{l-1,t} (layer below)
    if (tid == 0) {
      bool ok = wait_ge(cnt_own, (unsigned)(NCU * t), err, a.fail, a.spin_limit);
      if (!L0 && ok) ok = wait_ge(cnt_low, (unsigned)(NCU * (t + 1)), err, a.fail, a.spin_limit);
      PG_STAMP(1);
      // no agent acquire: published rows are only ever read with sc1 loads (pg_persist_bwd.hip, same place)
      if constexpr (AUXL == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *sFlagT = ok ? 1u : 0u;
    }
    const int o = sd.off[t], Bt = sd.off[t + 1] - o;
    const int op = t > 0 ? sd.off[t - 1] : 0;
    const int nlive = (Bt - g + NG - 1) / NG;          // sequences g, g+NG, ... below B_t
    const int ntile = (nlive + 15) >> 4;
    if constexpr (L0) {                                // this step's tokens of the group (nothing else is in flight here)
      if (tid < 128) sTok[tid] = a.tok[o + (tid < nlive ? g + NG * tid : g)];
    }
    __syncthreads();
    if (*sFlagT == 0u) return;   // block-uniform: somebody gave up, leave (every waiter does the same)
    const bf16_t* hprev_base = t == 0 ? a.h0_op : hseq + (long)op * H;
    const bf16_t* x_base = L0 ? hprev_base : xsrc + (long)o * H;
    const float* carry_src = (t == 0 ? a.h0_32 : carry) + c * 64;

    auto stage = [&](int j, int buf) __attribute__((always_inline)) {
      if (dbg & 1) return;
      char* tb = sT + buf * TILEB;
#pragma unroll
      for (int i = 0; i < RP_WAVE; ++i) {
        const int rp = wave * RP_WAVE + i;             // wave-uniform
        const int row = rp / PARTS, part = rp % PARTS;
        const int b = g + NG * (16 * j + row);
        const int bs = b < Bt ? b : g;                 // finished sequences re-read a valid row; their columns are unused
        const bf16_t* src = (part == 0 ? hprev_base : x_base) + (long)bs * H + ((lane ^ (row & 15)) * 8);
        char* dst = tb + row * (PARTS * 1024) + part * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, AUXL);
      }
      {  // epilogue inputs: this wave's quarter of the rows, lane = (row in quarter, 16-byte chunk of the 256-byte slice)
        const int row = wave * 4 + (lane >> 4), ch = (lane & 15) ^ row;
        const int i = 16 * j + row;
        const int b = g + NG * i;
        const int bs = b < Bt ? b : g;
        char* eb = tb + ACTB + wave * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(carry_src + (long)bs * H + ch * 4), (lptr_t)eb, 16, 0, 0);
        if constexpr (L0) {
          const float* tp = a.table + (long)sTok[i < 128 ? i : 0] * 3 * H + c * 64 + ch * 4;
          const float* zp = a.zproj + (long)bs * 3 * H + c * 64 + ch * 4;
#pragma unroll
          for (int gt = 0; gt < 3; ++gt) {
            __builtin_amdgcn_global_load_lds((gptr_t)(tp + gt * H), (lptr_t)(eb + G::CARRY_BYTES + gt * G::CARRY_BYTES), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(zp + gt * H), (lptr_t)(eb + G::CARRY_BYTES + G::GATE_BYTES + gt * G::CARRY_BYTES), 16, 0, 0);
          }
        }
      }
    };
    PG_STAMP(2);
    // both buffers are free at the start of a step: tiles 0 and 1 are requested together, the first wait leaves tile 1's
    // ND instructions in flight
    constexpr int ND = RP_WAVE + 1 + (L0 ? 6 : 0);     // DMA instructions per wave and tile
    static_assert(ND == 9 || ND == 11, "wait_all_but has no case for this tile");
    stage(0, 0);
    if (ntile > 1) stage(1, 1);                        // block-uniform
    if (dbg & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else wait_all_but(ntile > 1 ? ND : 0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PG_STAMP(3);
    // Order inside the loop: after the barrier that ends tile j come the DMA of tile j+2 and only then the stores of
    // tile j, so the wait at the end of tile j+1 ("tile j+2 has landed") leaves exactly those NS stores in flight: a
    // store is waited for two tiles after its issue at the earliest (its acknowledgement takes 2-3 us).  Nothing in the
    // loop loads from global memory into registers - hipcc would guard such a load with a wait for every older store.
    for (int j = 0; j < ntile; ++j) {
      const int buf = j & 1;
      const int b = g + NG * (16 * j + lr);
      const bool live = b < Bt;
      const char* tb = sT + buf * TILEB;
      // ---- gates: acc[r], acc[z] over [h | x]; acc[hn] over h; acc[in] over x
      f32x4 acc[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* A = tb + lr * (PARTS * 1024) + ((lq ^ lr) & 3) * 16;
      const int hi = lr >> 2;
      // LDS fragments are read in groups of 4 K-steps, two groups ahead of the MFMAs that consume them (the scheduling
      // barriers keep hipcc from sinking the reads back to their uses: one exposed LDS latency per MFMA otherwise)
      if (!(dbg & 2)) {
        bf16x8 f[KS / 4][4];
#pragma unroll
        for (int gk = 0; gk < 2; ++gk)
#pragma unroll
          for (int i = 0; i < 4; ++i) f[gk][i] = *reinterpret_cast<const bf16x8*>(A + (((gk * 4 + i) ^ hi) << 6));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gk = 0; gk < KS / 4; ++gk) {
          if (gk + 2 < KS / 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) f[gk + 2][i] = *reinterpret_cast<const bf16x8*>(A + ((((gk + 2) * 4 + i) ^ hi) << 6));
          }
          __builtin_amdgcn_sched_barrier(0);
          if (gk == 0) mma12_aaa<true>(acc[0], acc[1], acc[2], &wh[0][gk * 4], &wh[1][gk * 4], &wh[2][gk * 4], f[gk]);
          else mma12_aaa<DBG>(acc[0], acc[1], acc[2], &wh[0][gk * 4], &wh[1][gk * 4], &wh[2][gk * 4], f[gk]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (!L0) if (!(dbg & 2)) {
        const char* Wn = sWin + (wave * 16 + lr) * ROWW + ((lq ^ lr) & 3) * 16;   // chunk ks*4+lq of row lr sits at (ks*4+lq) ^ lr
        bf16x8 fx[KS / 4][4], fw[KS / 4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fx[0][i] = *reinterpret_cast<const bf16x8*>(A + 1024 + ((i ^ hi) << 6));
          fw[0][i] = *reinterpret_cast<const bf16x8*>(Wn + ((i ^ hi) << 6));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gk = 0; gk < KS / 4; ++gk) {
          if (gk + 1 < KS / 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int ks = (gk + 1) * 4 + i;
              fx[gk + 1][i] = *reinterpret_cast<const bf16x8*>(A + 1024 + ((ks ^ hi) << 6));
              fw[gk + 1][i] = *reinterpret_cast<const bf16x8*>(Wn + ((ks ^ hi) << 6));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (gk == 0) mma12_avv<true>(acc[0], acc[1], acc[3], &wi[0][gk * 4], &wi[1][gk * 4], fw[gk], fx[gk]);   // acc[3] starts here
          else mma12_avv<DBG>(acc[0], acc[1], acc[3], &wi[0][gk * 4], &wi[1][gk * 4], fw[gk], fx[gk]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      PG_MFMA_DRAIN(acc[0], acc[1], acc[2], acc[3]);
      // ---- epilogue on 4 units of sequence b; inputs from the tile's epilogue block (chunk q of row r sits at q ^ r)
      const char* E = tb + ACTB + lr * 256 + (((wave * 4 + lq) ^ lr) << 4);
      const f32x4 hp4 = *reinterpret_cast<const f32x4*>(E);
      f32x4 g3[3];
      if constexpr (L0) {
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
          g3[gt] = *reinterpret_cast<const f32x4*>(E + G::CARRY_BYTES + gt * G::CARRY_BYTES) +
                   *reinterpret_cast<const f32x4*>(E + G::CARRY_BYTES + G::GATE_BYTES + gt * G::CARRY_BYTES);
      } else {
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) g3[gt] = f32x4{bgi[gt][0], bgi[gt][1], bgi[gt][2], bgi[gt][3]};
      }
      float r[4], z[4], n[4], hn[4], hp[4], h[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hp[e] = hp4[e];
        r[e] = sigmoidf_(g3[0][e] + acc[0][e]);
        z[e] = sigmoidf_(g3[1][e] + acc[1][e]);
        hn[e] = acc[2][e] + bh[e];
        n[e] = tanhf_(g3[2][e] + acc[3][e] + r[e] * hn[e]);
        h[e] = (1.0f - z[e]) * n[e] + z[e] * hp[e];
      }
      // The rows this tile publishes or saves row-major go to memory through a small LDS staging block: a lane holds 4
      // units of one row, so a direct store is 16 rows x 32 bytes per wave instruction - 16 cache-line accesses that the
      // address path serialises (~35 CU cycles each) - while 128-byte row segments read back from LDS are 8 lines per
      // instruction and a third as many instructions.  Two buffers: the block is rewritten two barriers later.
      {
        char* S = sStage + (j & 1) * G::STG_BUF + lr * G::STG_ROW + (wave * 16 + lq * 4) * 2;
        *reinterpret_cast<bf16x4*>(S) = to_bf16x4(h);
        int k = 1;
        if (has_drop) {
          float hd[4], m[4];
          dropout_scale4(a.seed, l, t, b * a.b_mul + a.b_add, u0 >> 2, a.p_drop, inv_keep, m);
#pragma unroll
          for (int e = 0; e < 4; ++e) hd[e] = h[e] * m[e];
          *reinterpret_cast<bf16x4*>(S + k * G::STG_ARR) = to_bf16x4(hd);
          ++k;
        }
        if (has_sv) *reinterpret_cast<bf16x4*>(S + k * G::STG_ARR) = to_bf16x4(hp);
      }
      // tile j+1 must have landed (this wave's DMAs; the barrier extends it to the others): everything older than the
      // previous tile's stores.  The barrier also says that every wave is done reading tile j.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(dbg & 1)) wait_all_but(j > 0 ? NS : 0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (j + 2 < ntile) stage(j + 2, buf);            // block-uniform
      asm volatile("" ::: "memory");
      {
        const unsigned oob = 0x80000000u;              // beyond the descriptors' 2^31 - 1 records: the store is dropped
        const bool st = live && !((dbg & 4) && h[0] != 12345.678f);
        const long s = (long)(o + b) * H + u0;
        const unsigned o2 = st ? (unsigned)(s * 2) : oob;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{h[0], h[1], h[2], h[3]}), r_carry,
                                               (int)(st ? (unsigned)(((long)b * H + u0) * 4) : oob), 0, 0);
        // cooperative row-segment stores: 128 lanes per array (16 rows x 8 lanes of 16 bytes); waves 0-1 take the first
        // array of a round, waves 2-3 the second; every wave issues one store per round (out of range when it has none)
        const char* SB = sStage + (j & 1) * G::STG_BUF;
        const int row = (lane >> 3) + ((wave & 1) << 3), seg = lane & 7;
        const int brow = g + NG * (16 * j + row);
        const bool rlive = brow < Bt && !((dbg & 4) && h[0] != 12345.678f);
        const unsigned orow = (unsigned)((((long)(o + brow) * H + c * 64) * 2) + seg * 16);
        for (int rd = 0; rd < (narr > 2 ? 2 : 1); ++rd) {   // kernel-uniform trip count
          const int arr = 2 * rd + (wave >> 1);                // wave-uniform
          const u32x4 v = *reinterpret_cast<const u32x4*>(SB + (arr < 3 ? arr : 0) * G::STG_ARR + row * G::STG_ROW + seg * 16);
          const unsigned off = (arr < narr && rlive) ? orow : oob;
          if (arr == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r_hseq, (int)off, 0, AUXL);
          else if (arr == 1 && has_drop) __builtin_amdgcn_raw_buffer_store_b128(v, r_hdrop, (int)off, 0, AUXL);
          else __builtin_amdgcn_raw_buffer_store_b128(v, pg_make_rsrc(a.sv[l][4] ? a.sv[l][4] : hseq), (int)off, 0, 0);
        }
        if (has_sv) {
          // r, z, n, hn go back to the BPTT kernel lane for lane: stored in lane order they are 512 contiguous bytes per
          // wave instruction (pg_persist.h, sv_tile_slot); packed rows when the BPTT of this step is not the persistent kernel
          if (sv_lane) {   // kernel-uniform: (r, z) pairs in sv[l][0..1], (n, hn) pairs in sv[l][2..3], 16 bytes per lane
            const unsigned o16 = st ? (unsigned)(((sv_tile_slot(o, t, NG, j, g) * NCU + c) * 256 + wave * 64 + lane) * 16) : oob;
            const bf16x4 vr = to_bf16x4(r), vz = to_bf16x4(z), vn = to_bf16x4(n), vh = to_bf16x4(hn);
            const u32x2 ur = __builtin_bit_cast(u32x2, vr), uz = __builtin_bit_cast(u32x2, vz);
            const u32x2 un = __builtin_bit_cast(u32x2, vn), uh = __builtin_bit_cast(u32x2, vh);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{ur[0], ur[1], uz[0], uz[1]}, pg_make_rsrc(a.sv[l][0]), (int)o16, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{un[0], un[1], uh[0], uh[1]}, pg_make_rsrc(a.sv[l][2]), (int)o16, 0, 0);
          } else {
            store_pub_b64<0>(pg_make_rsrc(a.sv[l][0]), o2, to_bf16x4(r));
            store_pub_b64<0>(pg_make_rsrc(a.sv[l][1]), o2, to_bf16x4(z));
            store_pub_b64<0>(pg_make_rsrc(a.sv[l][2]), o2, to_bf16x4(n));
            store_pub_b64<0>(pg_make_rsrc(a.sv[l][3]), o2, to_bf16x4(hn));
          }
        }
      }
      if (j == 0) PG_STAMP(4);
    }
    PG_STAMP(5);
    // ---- publish step t of this unit slice
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PG_STAMP(6);
    if (tid == 0) {
      if constexpr (AUXL == 0) {   // plain payload stores: lane-0 agent release (L2 write-back) before the arrival
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(cnt_own, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PG_STAMP(7);
  }
}

template <int H, int AUXL, bool DBG, bool FAST>
__global__ __launch_bounds__(256) void dec_fwd_persist_kernel(DecFwdPersist a, Sched sd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int g, l, c;   // blockIdx % 8 labels the XCD: a group's workgroups (or at least a layer's) share one
  if (!persist_map(blockIdx.x, a.NG, DecGeo<H>::NCU, g, l, c)) return;
  if (l == 0) dec_fwd_body<H, 0, AUXL, DBG, FAST>(a, sd, g, l, c, smem);
  else if (l == a.L - 1) dec_fwd_body<H, 2, AUXL, DBG, FAST>(a, sd, g, l, c, smem);
  else dec_fwd_body<H, 1, AUXL, DBG, FAST>(a, sd, g, l, c, smem);
}

unsigned long long* g_stamps = nullptr;
int g_stamps_which = 0;            // 0: forward kernel, 1: BPTT kernel
int* g_fail_host = nullptr;        // pinned mirror of the sticky give-up word
unsigned* g_fail_dev = nullptr;   // sticky until persist_recover()
unsigned g_spin_limit = SPIN_LIMIT;

}  // namespace

// B <= 1024: a group owns at most 128 sequences (the token block in LDS of layer 0 is sized for that)
bool dec_persist_supported(int H, int L, int B) { return H == 512 && L >= 1 && L * (H / 64) * 8 <= 256 && B >= 1 && B <= 1024; }

int persist_poll_failure() { return g_fail_host ? *reinterpret_cast<volatile int*>(g_fail_host) : 0; }

// experiment switch PG_DMA_PLAIN=1: the LDS-DMA loads of published rows use the default cache policy (they sit behind the
// step's agent acquire, which makes plain loads a valid consumer form) instead of sc1
bool persist_dma_plain() {
  static const bool v = getenv("PG_DMA_PLAIN") && atoi(getenv("PG_DMA_PLAIN")) != 0;
  return v;
}

int persist_fail_words(unsigned** dev) {
  if (!g_fail_host) {
    PG_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&g_fail_host), 64, hipHostMallocDefault));
    *g_fail_host = 0;
    PG_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_fail_dev), 64));
    PG_CHECK_HIP(hipMemset(g_fail_dev, 0, 64));
  }
  *dev = g_fail_dev;
  return PG_OK;
}

unsigned* persist_fail_dev_or_null() { return g_fail_dev; }
unsigned persist_spin_limit() { return g_spin_limit; }
extern "C" __attribute__((visibility("default"))) unsigned pg_persist_spin_limit(long limit) {
  const unsigned old = g_spin_limit;
  if (limit >= 0) g_spin_limit = limit > 0xFFFFFFF0L ? 0xFFFFFFF0u : (unsigned)limit;
  return old;
}

int persist_recover() {
  if (!g_fail_host) return PG_OK;
  PG_CHECK_HIP(hipDeviceSynchronize());
  PG_CHECK_HIP(hipMemset(g_fail_dev, 0, 64));
  *g_fail_host = 0;
  return PG_OK;
}

int persist_fail_mirror(hipStream_t stream) {
  PG_CHECK_HIP(hipMemcpyAsync(g_fail_host, g_fail_dev, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
  return PG_OK;
}

int persist_fail_refresh() {
  if (!g_fail_host) return PG_OK;
  unsigned v = 0;
  PG_CHECK_HIP(hipMemcpy(&v, g_fail_dev, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (v) *g_fail_host = (int)v;
  return PG_OK;
}

int launch_dec_fwd_persist(DecFwdPersist& a, const Sched& sd, int H, hipStream_t stream, int host_flags) {
  PG_REQUIRE(dec_persist_supported(H, a.L, a.B), "dec_fwd_persist: shape H=%d L=%d not covered", H, a.L);
  PG_REQUIRE(a.sync && a.table && a.tok && a.zproj && a.h0_op && a.h0_32, "dec_fwd_persist: NULL buffer");
  using G = DecGeo<512>;
  a.NG = persist_groups(a.L, a.B);
  PG_TRY(persist_fail_words(&a.fail));
  a.spin_limit = persist_spin_limit();
  a.stamps = g_stamps_which == 0 ? g_stamps : nullptr;
  a.dbg = getenv("PG_PERSIST_DBG") ? atoi(getenv("PG_PERSIST_DBG")) : 0;
  if (!(host_flags & PG_PERSIST_SYNC_ZEROED))
    PG_CHECK_HIP(hipMemsetAsync(a.sync, 0, PG_PERSIST_SYNC_WORDS * sizeof(unsigned), stream));
  // (PG_FWD8_KO=32, diagnostic builds only: the stamps go to the two-waves-per-SIMD kernel's own instrumented instantiation)
  static const bool stamps8 = getenv("PG_FWD8_KO") && atoi(getenv("PG_FWD8_KO")) == 32;
  const bool dbgk = a.dbg != 0 || (a.stamps != nullptr && !stamps8);
  const bool plain = persist_dma_plain();
  bool fast = !dbgk && !plain && a.L >= 2 && a.sv_lane && !getenv("PG_PERSIST_NO_FAST");      // the training configuration (see dec_fwd_body)
  for (int l = 0; l < a.L; ++l) fast = fast && a.sv[l][0] && ((a.hdrop[l] != nullptr) == (l < a.L - 1));
  // Every workgroup waits for its peers, so the whole grid must be co-resident: one workgroup per CU (512 registers per
  // lane, 130-145 KB LDS) on a device with at least `grid` CUs.  Checked here; a plain launch has the same residency as a
  // cooperative one and works when several processes use the GPU (their kernels drain independently, waits are bounded).
  const int grid = a.NG == 8 ? 8 * a.L * G::NCU : 256;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    PG_CHECK_HIP(hipGetDevice(&dev));
    PG_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  PG_REQUIRE(grid <= n_cu, "dec_fwd_persist: %d workgroups cannot be co-resident on %d CUs", grid, n_cu);
  if (fast && dec_fwd8_enabled()) {   // the training configuration: two waves per SIMD (pg_persist_dec8.hip)
    PG_TRY(launch_dec_fwd8(a, sd, grid, stream));
    if (!(host_flags & PG_PERSIST_NO_MIRROR)) PG_TRY(persist_fail_mirror(stream));
    return PG_OK;
  }
  const void* kern = fast    ? reinterpret_cast<const void*>(dec_fwd_persist_kernel<512, AUX_SC1, false, true>)
                     : plain ? (dbgk ? reinterpret_cast<const void*>(dec_fwd_persist_kernel<512, 0, true, false>)
                                     : reinterpret_cast<const void*>(dec_fwd_persist_kernel<512, 0, false, false>))
                             : (dbgk ? reinterpret_cast<const void*>(dec_fwd_persist_kernel<512, AUX_SC1, true, false>)
                                     : reinterpret_cast<const void*>(dec_fwd_persist_kernel<512, AUX_SC1, false, false>));
  static bool attr_done[5] = {false, false, false, false, false};
  const int ki = fast ? 4 : plain * 2 + dbgk;
  if (!attr_done[ki]) {
    PG_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM));
    attr_done[ki] = true;
  }
  void* kargs[] = {&a, const_cast<Sched*>(&sd)};
  PG_CHECK_HIP(hipLaunchKernel(kern, dim3(grid), dim3(256), kargs, (size_t)G::SMEM, stream));
  PG_CHECK_LAUNCH();
  if (!(host_flags & PG_PERSIST_NO_MIRROR)) PG_TRY(persist_fail_mirror(stream));
  return PG_OK;
}

unsigned long long* persist_stamps_for(int which) { return g_stamps_which == which ? g_stamps : nullptr; }

// development aid (tools/persist_stamps.py): device buffer of 8 + 64 u64 words; word 0 = layer to trace;
// which = 0 traces the forward kernel, 1 the BPTT kernel
extern "C" __attribute__((visibility("default"))) int pg_persist_stamps(void* buf, int which) {
  g_stamps = static_cast<unsigned long long*>(buf);
  g_stamps_which = which;
  return PG_OK;
}
