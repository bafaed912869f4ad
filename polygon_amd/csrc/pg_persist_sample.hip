// Group-persistent autoregressive sampler (VAE.sample, vae_model.py:300-358): ONE launch decodes every step of every row.
//
// The per-step path (pg_sample.hip) launches 3 GRU cells + 1 head kernel per token and re-reads the decoder's 9.4 MB of
// weights from L2 in every launch.  Here the weights are stationary, exactly as in the training forward kernel
// (pg_persist_dec.hip): group g (one XCD) owns rows g, g+8, g+16, ...; workgroup (g, layer l, slice c) keeps the W_hh /
// W_ih rows of 64 hidden units in registers + LDS and streams 16-row tiles of [h_{l,t-1} | h_{l-1,t}] through LDS-DMA.
// What is new is the feedback: layer 0 of step t+1 needs the token drawn from the top layer's output of step t.  A fourth
// stage per group, the HEAD workgroups (8 per group, a tile each in turn), closes the loop: vocabulary projection on MFMA
// (W_fc stationary in registers), softmax and inverse-CDF draw (the arithmetic of sample_head_kernel, operation for
// operation), eos bookkeeping, and the token is published for layer 0.
// Rows are independent, so a group splits its rows into S SUB-BATCHES that travel round the 4-stage ring
// l0 -> l1 -> l2 -> head -> l0 one behind the other: with S >= 4 every stage has a sub-batch to work on while the others
// are elsewhere.  Hand-off: one arrival counter per (group, stage, sub-batch); stage k starts (t, s) when stage k-1 has
// completed (t, s) and all slices of stage k have completed (t-1, s); layer 0 waits for the head's (t-1, s).  Published
// rows are write-through stores read with sc1 loads; every wait is bounded (pg_persist_dec.hip, protocol).
// h_{l,t} ping-pongs between two [n, H] buffers per layer: h_{l,t+1} may overwrite h_{l,t-1} because the token feedback
// orders (l, t+1, s) behind the head's (t, s), hence behind every reader of h_{l,t-1}.
#include <hip/hip_runtime.h>
#include "pg_mma.h"
#include "pg_persist.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

namespace {

constexpr int H = 512, KS = H / 32, NCU = H / 64, NG = 8, NH = 8, NSTAGE = 4;
constexpr int SYNC_STRIDE = 16;
constexpr int SYNC_ERR = NG * NSTAGE * PG_SAMPLE_MAX_SUB * SYNC_STRIDE;
static_assert(SYNC_ERR + 1 <= PG_SAMPLE_SYNC_WORDS, "sync block too small");
constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr int AUX_SC1 = 16;
constexpr int ROWW = H * 2;   // W_in row in LDS: unpadded, 16-byte chunk q of row u at position q ^ (u & 15)
constexpr int WIN_BYTES = 64 * ROWW;
constexpr int ACT_BYTES = 16 * 2 * H * 2;
constexpr int CARRY_BYTES = 16 * 64 * 4;
constexpr int GATE_BYTES = 3 * CARRY_BYTES;
constexpr int TILE_BYTES = ACT_BYTES + CARRY_BYTES;
constexpr int TILE0_BYTES = ACT_BYTES / 2 + CARRY_BYTES + 2 * GATE_BYTES;
constexpr int TOK_BYTES = 16 * PG_SAMPLE_MAX_TS * 4;
constexpr int LIST_BYTES = 16 * PG_SAMPLE_MAX_TS * 4 + 16;   // live-row list of the visit + the four waves' counts
constexpr int SMEM = WIN_BYTES + 2 * TILE_BYTES + LIST_BYTES + 16;
constexpr int LIST_OFF = SMEM - 16 - LIST_BYTES;
static_assert(2 * TILE0_BYTES + TOK_BYTES <= LIST_OFF, "layer-0 layout must fit the common allocation");
static_assert(2 * 16 * 1024 + 16 * 65 * 4 <= LIST_OFF, "head layout must fit the common allocation");
static_assert(16 * PG_SAMPLE_MAX_TS <= 256, "one thread per row of a sub-batch builds the live list");
static_assert(PG_SAMPLE_MAX_TS <= 2 * NH, "a head workgroup stages at most two tiles per visit");

__device__ __forceinline__ f32x4 mma16(const bf16x8& w, const bf16x8& x, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
}
// weight operand straight from an AGPR; see pg_persist_dec.hip for the hazard rules
__device__ __forceinline__ f32x4 mma16a(const bf16x8& w, const bf16x8& x, f32x4 c) {
  asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(w), "v"(x));
  return c;
}
#define PG_MFMA_DRAIN(a0, a1, a2, a3) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3))
__device__ __forceinline__ bf16x4 to_bf16x4(const float (&v)[4]) {
  bf16x4 o;
  o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
  return o;
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

__device__ __forceinline__ unsigned* counter(const DecSamplePersist& a, int g, int stage, int s) {
  return a.sync + ((g * NSTAGE + stage) * PG_SAMPLE_MAX_SUB + s) * SYNC_STRIDE;
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Live rows of sub-batch [j0, j1) of group g at step t, compacted in row order: sList[k] = index within the sub-batch
// of the k-th row that is still decoding, padded with -1 to a multiple of 16; sTokC[k] (layer 0) = the token that row
// feeds.  tok[b] >= 0 is the token of a decoding row; a row whose eos was drawn at step te carries -1 - te and stays in
// the lists up to and including step te + 1 (that last step is wasted: the head records nothing for it).  The extra
// step buys this: the words that decide the list of visit (t, s) are those the head published up to step t - 2 -- all
// complete once ANY stage has finished its (t-1, s) -- while everything published later (a token, -1 - (t-1), -1 - t)
// reads as "live at t".  So every stage may load the words whenever it likes after its own (t-1, s), concurrently with
// the head's writes, and all of them still build the same list: layers 1, 2 load them BEFORE they start to wait for
// the stage below, which takes the L2 round trip off the ring's critical path; layer 0 and the head load them behind
// their wait, because they need the newest value too (the token to feed; whether the row ended at t - 1).
// Ended rows cost nothing from the second step on.  A row's arithmetic does not depend on its tile slot and its
// uniforms are keyed by the row, so the tokens are those of the uncompacted decode.
constexpr int NOT_A_ROW = -0x7fffffff - 1;
__device__ __forceinline__ int load_tok_word(const DecSamplePersist& a, int g, int j0, int j1) {
  const int tid = threadIdx.x;
  const int b = g + NG * (16 * j0 + tid);
  int tv = NOT_A_ROW;
  if (tid < 16 * (j1 - j0) && b < a.n) tv = __hip_atomic_load(a.tok + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return tv;
}
// HEAD: entries carry 256 * (the row ended at t - 1) on top of the index.  Returns the live count.
template <bool HEAD>
__device__ __forceinline__ int build_live_list(int tv, int t, char* smem, int* sTokC) {
  int* sList = reinterpret_cast<int*>(smem + LIST_OFF);
  int* sCnt = sList + 16 * PG_SAMPLE_MAX_TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool live = tv >= 0 || (tv != NOT_A_ROW && -1 - tv >= t - 1);
  const unsigned long long m = __ballot(live);
  const int pre = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) sCnt[wave] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int cw = sCnt[w];
    off += w < wave ? cw : 0;
    tot += cw;
  }
  if (live) {
    sList[off + pre] = HEAD ? tid + (tv < 0 && -1 - tv < t ? 256 : 0) : tid;
    if (sTokC) sTokC[off + pre] = tv >= 0 ? tv : 0;
  }
  if (tid >= tot && tid < ((tot + 15) & ~15)) {
    sList[tid] = -1;
    if (sTokC) sTokC[tid] = 0;
  }
  __syncthreads();
  return tot;
}

// one GRU layer of the group: the tile loop of dec_fwd_body (pg_persist_dec.hip) without the training outputs
template <bool L0>
__device__ __forceinline__ void layer_body(const DecSamplePersist& a, int g, int l, int c, char* smem) {
  char* sWin = smem;
  char* sA = smem + WIN_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int uw = c * 64 + wave * 16;
  const int u0 = uw + lq * 4;

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
  float* carry = a.carry[l];
  const __amdgpu_buffer_rsrc_t r_carry = pg_make_rsrc(carry);
  unsigned* err = a.sync + SYNC_ERR;
  __syncthreads();

  constexpr int PARTS = L0 ? 1 : 2;
  constexpr int RP_WAVE = 16 * PARTS / 4;
  constexpr int TILEB = L0 ? TILE0_BYTES : TILE_BYTES;
  constexpr int ACTB = ACT_BYTES / (L0 ? 2 : 1);
  char* sT = L0 ? smem : sA;
  int* sTok = reinterpret_cast<int*>(smem + 2 * TILE0_BYTES);      // layer 0 only
  volatile unsigned* sFlagT = reinterpret_cast<volatile unsigned*>(smem + SMEM - 16);
  const int* sList = reinterpret_cast<const int*>(smem + LIST_OFF);
  constexpr int NS = 2;                                            // stores per wave and tile: state, published h
  constexpr int ND = RP_WAVE + 1 + (L0 ? 6 : 0);                   // DMA instructions per wave and tile
  static_assert(ND == 9 || ND == 11, "unexpected tile shape");
  const float* carry_src = carry + c * 64;

  for (int t = 0; t < a.T; ++t) {
    const int p = t & 1;
    const bf16_t* hprev_base = a.hop[l][p];
    const bf16_t* x_base = L0 ? hprev_base : a.hop[l - 1][p ^ 1];
    const __amdgpu_buffer_rsrc_t r_h = pg_make_rsrc(a.hop[l][p ^ 1]);
    int live_t = 0;
    for (int s = 0; s < a.S; ++s) {
      const int j0 = s * a.TS;
      const int j1 = j0 + a.TS < a.ntile ? j0 + a.TS : a.ntile;
      int tv = 0;
      if constexpr (!L0) tv = load_tok_word(a, g, j0, j1);   // in flight while thread 0 polls
      // ---- wait for (l, t-1, s) of every slice and for the stage below: (l-1, t, s), or the head's (t-1, s)
      if (tid == 0) {
        bool ok = wait_ge(counter(a, g, l, s), (unsigned)(NCU * t), err, a.fail, a.spin_limit);
        if (ok) {
          if constexpr (L0) ok = wait_ge(counter(a, g, NSTAGE - 1, s), (unsigned)(NH * t), err, a.fail, a.spin_limit);
          else ok = wait_ge(counter(a, g, l - 1, s), (unsigned)(NCU * (t + 1)), err, a.fail, a.spin_limit);
        }
        // no agent acquire: published rows and tokens are only read with sc1 / agent-scope loads (pg_persist_bwd.hip)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *sFlagT = ok ? 1u : 0u;
      }
      __syncthreads();
      if (*sFlagT == 0u) return;   // block-uniform: somebody gave up
      // the sub-batch's live rows and (layer 0) their tokens, drawn by the head workgroups in this launch: coherent loads
      if constexpr (L0) tv = load_tok_word(a, g, j0, j1);
      const int nlive = build_live_list<false>(tv, t, smem, L0 ? sTok : nullptr);
      const int nt = (nlive + 15) >> 4;   // tiles of this visit, block-uniform
      live_t += nlive;
      // global row of a list entry, -1: empty slot.  A tile's entries are read from LDS two tiles ahead (ent_*), at the
      // top of the iteration that ends by staging it, so their latency is off the tile's critical path: er = the entry
      // of row lane & 15 (the DMA rows of a wave come out of it by readlane), ec = the entry of the carry row
      auto dec_row = [&](int li) __attribute__((always_inline)) { return li >= 0 ? g + NG * (16 * j0 + li) : -1; };
      auto ent_r = [&](int jj) __attribute__((always_inline)) { return jj < nt ? sList[16 * jj + lr] : -1; };
      auto ent_c = [&](int jj) __attribute__((always_inline)) { return jj < nt ? sList[16 * jj + wave * 4 + lq] : -1; };
      auto ent_t = [&](int jj) __attribute__((always_inline)) { return L0 && jj < nt ? sTok[16 * jj + wave * 4 + lq] : 0; };

      auto stage = [&](int buf, int er, int ec, int tk) __attribute__((always_inline)) {
        char* tb = sT + buf * TILEB;
#pragma unroll
        for (int i = 0; i < RP_WAVE; ++i) {
          const int rp = wave * RP_WAVE + i;
          const int row = rp / PARTS, part = rp % PARTS;
          const int b = dec_row(__builtin_amdgcn_readlane(er, row));
          const int bs = b >= 0 ? b : g;
          const bf16_t* src = (part == 0 ? hprev_base : x_base) + (long)bs * H + ((lane ^ (row & 15)) * 8);
          char* dst = tb + row * (PARTS * 1024) + part * 1024;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, AUX_SC1);
        }
        {
          const int row = wave * 4 + (lane >> 4), ch = (lane & 15) ^ row;
          const int b = dec_row(ec);
          const int bs = b >= 0 ? b : g;
          char* eb = tb + ACTB + wave * 1024;
          __builtin_amdgcn_global_load_lds((gptr_t)(carry_src + (long)bs * H + ch * 4), (lptr_t)eb, 16, 0, 0);
          if constexpr (L0) {
            const float* tp = a.table + (long)tk * 3 * H + c * 64 + ch * 4;
            const float* zp = a.zproj + (long)bs * 3 * H + c * 64 + ch * 4;
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) {
              __builtin_amdgcn_global_load_lds((gptr_t)(tp + gt * H), (lptr_t)(eb + CARRY_BYTES + gt * CARRY_BYTES), 16, 0, 0);
              __builtin_amdgcn_global_load_lds((gptr_t)(zp + gt * H), (lptr_t)(eb + CARRY_BYTES + GATE_BYTES + gt * CARRY_BYTES), 16, 0, 0);
            }
          }
        }
      };
      int er0 = ent_r(0), er1 = ent_r(1);
      if (nt > 0) stage(0, er0, ent_c(0), ent_t(0));
      if (nt > 1) {                // block-uniform
        stage(1, er1, ent_c(1), ent_t(1));
        if constexpr (ND == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int j = 0; j < nt; ++j) {
        const int buf = j & 1;
        const int er2 = ent_r(j + 2), ec2 = ent_c(j + 2), tk2 = ent_t(j + 2);
        const int b = dec_row(er0);
        const bool live = b >= 0;
        const char* tb = sT + buf * TILEB;
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* A = tb + lr * (PARTS * 1024) + ((lq ^ lr) & 3) * 16;
        const int hi = lr >> 2;
        {
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
            else mma12_aaa<false>(acc[0], acc[1], acc[2], &wh[0][gk * 4], &wh[1][gk * 4], &wh[2][gk * 4], f[gk]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if constexpr (!L0) {
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
            else mma12_avv<false>(acc[0], acc[1], acc[3], &wi[0][gk * 4], &wi[1][gk * 4], fw[gk], fx[gk]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        PG_MFMA_DRAIN(acc[0], acc[1], acc[2], acc[3]);
        const char* E = tb + ACTB + lr * 256 + (((wave * 4 + lq) ^ lr) << 4);
        const f32x4 hp4 = *reinterpret_cast<const f32x4*>(E);
        f32x4 g3[3];
        if constexpr (L0) {
#pragma unroll
          for (int gt = 0; gt < 3; ++gt)
            g3[gt] = *reinterpret_cast<const f32x4*>(E + CARRY_BYTES + gt * CARRY_BYTES) +
                     *reinterpret_cast<const f32x4*>(E + CARRY_BYTES + GATE_BYTES + gt * CARRY_BYTES);
        } else {
#pragma unroll
          for (int gt = 0; gt < 3; ++gt) g3[gt] = f32x4{bgi[gt][0], bgi[gt][1], bgi[gt][2], bgi[gt][3]};
        }
        float h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r = sigmoidf_(g3[0][e] + acc[0][e]);
          const float z = sigmoidf_(g3[1][e] + acc[1][e]);
          const float hn = acc[2][e] + bh[e];
          const float nn = tanhf_(g3[2][e] + acc[3][e] + r * hn);
          h[e] = (1.0f - z) * nn + z * hp4[e];
        }
        // tile j+1 has landed: everything older than the previous tile's NS stores (see pg_persist_dec.hip)
        if (j > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_assert(NS == 2, "the wait above counts two stores per tile");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + 2 < nt) stage(buf, er2, ec2, tk2);   // block-uniform
        asm volatile("" ::: "memory");
        {
          const unsigned oob = 0x80000000u;          // beyond the descriptor's records: the store is dropped
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{h[0], h[1], h[2], h[3]}), r_carry,
                                                 (int)(live ? (unsigned)(((long)b * H + u0) * 4) : oob), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, to_bf16x4(h)), r_h,
                                                (int)(live ? (unsigned)(((long)b * H + u0) * 2) : oob), 0, AUX_SC1);
        }
        er0 = er1; er1 = er2;
      }
      // ---- publish (t, s) of this unit slice
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(counter(a, g, l, s), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (live_t == 0) return;       // every row of the group has ended: all its stages see that at this same step
  }
}

// head workgroup hc of group g: tiles j0 + hc and j0 + hc + NH of every sub-batch (TS <= 2 * NH)
__device__ __forceinline__ void head_body(const DecSamplePersist& a, int g, int hc, char* smem) {
  char* sA = smem;                                                 // 2 x [16 rows][1 KB], chunk q of row r at q ^ r
  float* sl = reinterpret_cast<float*>(smem + 2 * 16 * 1024);      // [16][65] logits, then exp, then probabilities
  volatile unsigned* sFlagT = reinterpret_cast<volatile unsigned*>(smem + SMEM - 16);
  const int* sList = reinterpret_cast<const int*>(smem + LIST_OFF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int V = a.V, top = a.L - 1;
  bf16x8 wf[KS];                                                   // W_fc rows 16*wave + lr (padded to 64 rows)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    wf[ks] = *reinterpret_cast<const bf16x8*>(a.fcw + (long)(wave * 16 + lr) * H + (ks * 4 + lq) * 8);
  float fb[4];                                                     // bias of this lane's 4 vocabulary entries
#pragma unroll
  for (int e = 0; e < 4; ++e) fb[e] = a.fcb[wave * 16 + lq * 4 + e];
  const float inv_temp = a.inv_temp;
  unsigned* err = a.sync + SYNC_ERR;
  // softmax / draw mapping: thread = (row tid / 16, entries 4 * (tid % 16) .. + 3); the 16 threads of a row sit in one wave
  const int srow = tid >> 4, sq = tid & 15;
  float* y = sl + srow * 65;

  for (int t = 0; t < a.T; ++t) {
    const int step = t + 1;                                        // column of the token matrix this step fills
    const bf16_t* htop = a.hop[top][(t & 1) ^ 1];
    int live_t = 0;
    for (int s = 0; s < a.S; ++s) {
      const int j0 = s * a.TS;
      const int j1 = j0 + a.TS < a.ntile ? j0 + a.TS : a.ntile;
      if (tid == 0) {
        const bool ok = wait_ge(counter(a, g, top, s), (unsigned)(NCU * (t + 1)), err, a.fail, a.spin_limit);
        // no agent acquire: published rows and tokens are only read with sc1 / agent-scope loads (pg_persist_bwd.hip)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *sFlagT = ok ? 1u : 0u;
      }
      __syncthreads();
      if (*sFlagT == 0u) return;
      const int nlive = build_live_list<true>(load_tok_word(a, g, j0, j1), t, smem, nullptr);
      const int nt = (nlive + 15) >> 4;
      live_t += nlive;
      auto row_b = [&](int jj, int row) __attribute__((always_inline)) {
        const int li = sList[16 * jj + row];
        return li >= 0 ? g + NG * (16 * j0 + (li & 255)) : -1;
      };
      // both tiles of the visit are requested at once
      for (int j = hc, q = 0; j < nt; j += NH, ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wave * 4 + i;
          const int b = row_b(j, row);
          const int bs = b >= 0 ? b : g;
          __builtin_amdgcn_global_load_lds((gptr_t)(htop + (long)bs * H + ((lane ^ row) * 8)),
                                           (lptr_t)(sA + q * 16 * 1024 + row * 1024), 16, 0, AUX_SC1);
        }
      }
      for (int j = hc, q = 0; j < nt; j += NH, ++q) {
        if (q == 0 && j + NH < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // block-uniform
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* A = sA + q * 16 * 1024 + lr * 1024 + ((lq ^ lr) & 3) * 16;
        const int hi = lr >> 2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mma16(wf[ks], *reinterpret_cast<const bf16x8*>(A + ((ks ^ hi) << 6)), acc);
        // x = (logit + bias) / temp, the arithmetic of sample_head_kernel; entries >= V never win
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int col = wave * 16 + lq * 4 + e;
          sl[lr * 65 + col] = col < V ? (acc[e] + fb[e]) * inv_temp : -INFINITY;
        }
        __syncthreads();
        // max and first arg-max of the row (exact in any order), then e = exp(x - max) per entry
        float x[4], mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = y[sq * 4 + e]; mx = fmaxf(mx, x[e]); }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        int amax = 1 << 20;
#pragma unroll
        for (int e = 3; e >= 0; --e) if (x[e] == mx) amax = sq * 4 + e;
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) amax = min(amax, __shfl_xor(amax, o, 16));
        int w = amax;
        if (a.multinomial) {                                       // kernel-uniform
          float ex[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            ex[e] = sq * 4 + e < V ? __expf(x[e] - mx) : 0.0f;
            y[sq * 4 + e] = ex[e];
          }
          __syncthreads();
          // the sum runs in vocabulary order on one lane per row (fp32 addition is not associative: the per-step kernel's
          // order is kept, so both paths draw the same token from the same logits)
          float sum = 0.0f;
          if (sq == 0) for (int v = 0; v < V; ++v) sum += y[v];
          sum = __shfl(sum, 0, 16);
          __syncthreads();
#pragma unroll
          for (int e = 0; e < 4; ++e) y[sq * 4 + e] = ex[e] / sum;
          if (a.probs_out) {                                       // parity hook: the probabilities the draw below sees
            const int b = row_b(j, srow);
            if (b >= 0)
              *reinterpret_cast<f32x4*>(a.probs_out + ((long)(step - 1) * a.p_stride + b) * 64 + sq * 4) =
                  f32x4{ex[0] / sum, ex[1] / sum, ex[2] / sum, ex[3] / sum};
          }
          __syncthreads();
          if (sq == 0) {
            const int b = row_b(j, srow);
            float u = 2.0f;
            if (b >= 0) {
              if (a.uniforms) u = a.uniforms[(long)(step - 1) * a.u_stride + b];
              else {
                const unsigned long long gr = (unsigned long long)(a.row_offset + b);
                Philox4 r = philox4x32_10((uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)step, 0x5A17u, (uint32_t)a.seed,
                                          (uint32_t)(a.seed >> 32));
                u = pg_u01(r.x);
              }
            }
            float cdf = 0.0f;
            w = V - 1;
            bool found = false;
            for (int v = 0; v < V; ++v) {                          // first v with cdf_v > u
              cdf += y[v];
              if (!found && cdf > u) { w = v; found = true; }
            }
          }
        }
        if (sq == 0) {
          const int li = sList[16 * j + srow];
          if (li >= 0 && li < 256) {                               // a row, and not one that ended a step ago
            const int b = g + NG * (16 * j0 + li);
            int fed = a.forced ? (int)a.forced[(long)b * a.max_len + step] : w;      // parity hook: teacher forcing
            // eos[b]: written and read by one workgroup when compact == 0 (the list never changes); otherwise a row in
            // the list with li < 256 has not ended, whichever workgroup serves it now
            if (!a.eos[b]) {
              a.tokens[(long)b * a.max_len + step] = (uint8_t)w;
              if (w == a.eos_id) {
                a.end_pads[b] = step + 1;
                a.eos[b] = 1;
                // an ended row leaves the decode (nothing it would still compute is ever recorded); the teacher-forced
                // hook keeps every row to the last step: its probability dump is compared along the forced stream
                if (a.compact) fed = -1 - t;
              }
            }
            __hip_atomic_store(a.tok + b, fed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // published for layer 0
          }
        }
        __syncthreads();                                           // sl is reused by the next tile
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(counter(a, g, NSTAGE - 1, s), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (live_t == 0) return;       // the layers leave at this same step (build_live_list)
  }
}

__global__ __launch_bounds__(256) void dec_sample_persist_kernel(DecSamplePersist a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int g = blockIdx.x % NG;              // blockIdx % 8 labels the XCD: a group lives on one
  const int k = blockIdx.x / NG;
  if (k < NCU) layer_body<true>(a, g, 0, k, smem);
  else if (k < 3 * NCU) layer_body<false>(a, g, k / NCU, k % NCU, smem);
  else head_body(a, g, k - 3 * NCU, smem);
}

}  // namespace

static int device_cus() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = -1;
  }
  return n_cu;
}

// every workgroup waits for its peers: the grid must be co-resident, one workgroup per CU (pg_persist_dec.hip)
bool dec_sample_persist_supported(int Hd, int L, int V, long n) {
  return Hd == H && L == 3 && V <= 64 && n >= 1 && n <= (long)NG * 16 * PG_SAMPLE_MAX_SUB * PG_SAMPLE_MAX_TS &&
         device_cus() >= NG * (3 * NCU + NH);
}

int launch_dec_sample_persist(DecSamplePersist& a, hipStream_t stream) {
  PG_REQUIRE(dec_sample_persist_supported(H, a.L, a.V, a.n), "dec_sample_persist: shape L=%d V=%d n=%d not covered", a.L, a.V, a.n);
  PG_REQUIRE(a.sync && a.table && a.tok && a.eos && a.zproj && a.fcw && a.fcb && a.tokens && a.end_pads, "dec_sample_persist: NULL buffer");
  PG_REQUIRE(a.T >= 1 && a.max_len == a.T + 1, "dec_sample_persist: bad step count");
  a.ntile = (int)((((long)a.n + NG - 1) / NG + 15) / 16);
  // Sub-batches: 4 keep the 4 stages of a group busy; more only add visits, and a visit costs every stage its fixed
  // hand-off (2 us in a layer, ~9 us of latency through the head).  Measured at n = 256 .. 16384: TS = ntile/4 wins
  // (ntile/2 below 5 tiles, where a visit is all latency); PG_SAMPLE_TS overrides for experiments.
  int TS = a.ntile <= 4 ? (a.ntile + 1) / 2 : (a.ntile + 3) / 4;
  if (getenv("PG_SAMPLE_TS")) TS = atoi(getenv("PG_SAMPLE_TS"));
  const int ts_min = (a.ntile + PG_SAMPLE_MAX_SUB - 1) / PG_SAMPLE_MAX_SUB;
  if (TS < ts_min) TS = ts_min;
  if (TS > PG_SAMPLE_MAX_TS) TS = PG_SAMPLE_MAX_TS;
  if (TS > a.ntile) TS = a.ntile;
  if (TS < 1) TS = 1;
  a.TS = TS;
  a.S = (a.ntile + TS - 1) / TS;
  PG_REQUIRE(a.S >= 1 && a.S <= PG_SAMPLE_MAX_SUB, "dec_sample_persist: %d sub-batches", a.S);
  PG_TRY(persist_fail_words(&a.fail));
  a.spin_limit = persist_spin_limit();
  PG_CHECK_HIP(hipMemsetAsync(a.sync, 0, PG_SAMPLE_SYNC_WORDS * sizeof(unsigned), stream));
  static bool attr_done = false;
  if (!attr_done) {
    PG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dec_sample_persist_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_done = true;
  }
  const int grid = NG * (3 * NCU + NH);
  hipLaunchKernelGGL(dec_sample_persist_kernel, dim3(grid), dim3(256), (size_t)SMEM, stream, a);
  PG_CHECK_LAUNCH();
  PG_TRY(persist_fail_mirror(stream));
  return PG_OK;
}
