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
constexpr int SYNC_ERR = PG_MAX_LAYERS * 8 * SYNC_STRIDE;         // give-up word, after the [8][PG_MAX_LAYERS] counters
static_assert(SYNC_ERR + 1 <= PG_PERSIST_SYNC_WORDS, "sync block too small");
constexpr unsigned SPIN_LIMIT = 1u << 21;                         // ~2-4 s of polling before a workgroup gives up
constexpr int AUX_SC1 = 16;

__device__ __forceinline__ f32x4 mma16(const bf16x8& w, const bf16x8& x, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 to_bf16x4(const float (&v)[4]) {
  bf16x4 o;
  o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
  return o;
}
// 8-byte write-through store (published data)
__device__ __forceinline__ void store_sc1_b64(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const bf16x4& v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)byte_off, 0, AUX_SC1);
}
__device__ __forceinline__ uint4 load_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX_SC1);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// one lane: wait until *p >= target.  false: this or another workgroup gave up (err word set).
__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned target, unsigned* err, unsigned* sticky) {
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if ((spins & 63u) == 63u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    if (spins > SPIN_LIMIT) {
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
    if (a.stamps && tid == 0 && g == 0 && c == 0 && l == (int)a.stamps[0] && t >= 8 && t < 16)           \
      a.stamps[8 + (t - 8) * 8 + (slot)] = wall_clock64();                                               \
  } while (0)

template <int H> struct DecGeo {
  static constexpr int KS = H / 32;             // MFMA K steps over a K = H contraction
  static constexpr int NCU = H / 64;            // workgroups (unit slices) per layer
  static constexpr int ROWW = H * 2 + 16;       // W_in row in LDS (padded)
  static constexpr int WIN_BYTES = 64 * ROWW;
  static constexpr int TILE_BYTES = 16 * 2 * H * 2;   // activation tile rows [h_{t-1} | x], unpadded (swizzled)
  static constexpr int SMEM = WIN_BYTES + 2 * TILE_BYTES + 16;   // + the broadcast word of the poll result
};

template <int H, bool L0>
__device__ __forceinline__ void dec_fwd_body(const DecFwdPersist& a, const Sched& sd, int g, int l, int c, char* smem) {
  using G = DecGeo<H>;
  constexpr int KS = G::KS, NCU = G::NCU, ROWW = G::ROWW;
  static_assert(H == 512, "one LDS-DMA instruction (64 lanes x 16 B) must cover exactly one row half");
  char* sWin = smem;
  char* sA = smem + G::WIN_BYTES;
  volatile unsigned* sFlag = reinterpret_cast<volatile unsigned*>(smem + G::WIN_BYTES + 2 * G::TILE_BYTES);
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
      *reinterpret_cast<uint4*>(sWin + u * ROWW + ch * 16) =
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
  constexpr int PARTS = L0 ? 1 : 2;                 // row halves per row
  constexpr int RP_WAVE = 16 * PARTS / 4;           // (row, part) pairs per wave per tile
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  for (int t = 0; t < Tg; ++t) {
    PG_STAMP(0);
    // ---- wait for h_{l,t-1} (own layer) and x = h_{l-1,t} (layer below)
    if (tid == 0) {
      bool ok = wait_ge(cnt_own, (unsigned)(NCU * t), err, a.fail);
      if (!L0 && ok) ok = wait_ge(cnt_low, (unsigned)(NCU * (t + 1)), err, a.fail);
      PG_STAMP(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *sFlag = ok ? 1u : 0u;
    }
    __syncthreads();
    if (*sFlag == 0u) return;   // block-uniform: somebody gave up, leave (every waiter does the same)
    const int o = sd.off[t], Bt = sd.off[t + 1] - o;
    const int op = t > 0 ? sd.off[t - 1] : 0;
    const int nlive = (Bt - g + NG - 1) / NG;          // sequences g, g+NG, ... below B_t
    const int ntile = (nlive + 15) >> 4;
    const bf16_t* hprev_base = t == 0 ? a.h0_op : hseq + (long)op * H;
    const bf16_t* x_base = L0 ? hprev_base : xsrc + (long)o * H;

    auto stage = [&](int j, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < RP_WAVE; ++i) {
        const int rp = wave * RP_WAVE + i;             // wave-uniform
        const int row = rp / PARTS, part = rp % PARTS;
        const int b = g + NG * (16 * j + row);
        const int bs = b < Bt ? b : g;                 // finished sequences re-read a valid row; their columns are unused
        const bf16_t* src = (part == 0 ? hprev_base : x_base) + (long)bs * H + ((lane ^ (row & 15)) * 8);
        char* dst = sA + buf * G::TILE_BYTES + row * (PARTS * 1024) + part * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, AUX_SC1);
      }
    };
    PG_STAMP(2);
    stage(0, 0);
    int tok_next = 0;
    if constexpr (L0) {
      const int b0 = g + NG * lr;
      tok_next = a.tok[o + (b0 < Bt ? b0 : g)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PG_STAMP(3);
    for (int j = 0; j < ntile; ++j) {
      const int buf = j & 1;
      // epilogue inputs of this lane's sequence first, then the DMA of the next tile: the in-order vmcnt wait for the
      // former must not drag the latter along
      const int b = g + NG * (16 * j + lr);
      const bool live = b < Bt;
      const int bs = live ? b : g;                     // a valid row for the loads of finished sequences
      const f32x4 hp4 = *reinterpret_cast<const f32x4*>((t == 0 ? a.h0_32 : carry) + (long)bs * H + u0);
      f32x4 gi[3], gz[3];
      if constexpr (L0) {
        const float* tp = a.table + (long)tok_next * 3 * H + u0;
        const float* zp = a.zproj + (long)bs * 3 * H + u0;
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
          gi[gt] = *reinterpret_cast<const f32x4*>(tp + gt * H);
          gz[gt] = *reinterpret_cast<const f32x4*>(zp + gt * H);
        }
        if (j + 1 < ntile) {
          const int bn = g + NG * (16 * (j + 1) + lr);
          tok_next = a.tok[o + (bn < Bt ? bn : g)];
        }
      } else {
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
          gi[gt] = f32x4{bgi[gt][0], bgi[gt][1], bgi[gt][2], bgi[gt][3]};
          gz[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      if (j + 1 < ntile) stage(j + 1, buf ^ 1);        // block-uniform
      // ---- gates: acc[r], acc[z] over [h | x]; acc[hn] over h; acc[in] over x
      f32x4 acc[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* A = sA + buf * G::TILE_BYTES + lr * (PARTS * 1024) + ((lq ^ lr) & 3) * 16;
      const int hi = lr >> 2;
      {
        bf16x8 xc = *reinterpret_cast<const bf16x8*>(A + ((0 ^ hi) << 6)), xn = xc;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (ks + 1 < KS) xn = *reinterpret_cast<const bf16x8*>(A + (((ks + 1) ^ hi) << 6));
          acc[0] = mma16(wh[0][ks], xc, acc[0]);
          acc[1] = mma16(wh[1][ks], xc, acc[1]);
          acc[2] = mma16(wh[2][ks], xc, acc[2]);
          xc = xn;
        }
      }
      if constexpr (!L0) {
        const char* Wn = sWin + (wave * 16 + lr) * ROWW + lq * 16;
        bf16x8 xc = *reinterpret_cast<const bf16x8*>(A + 1024 + ((0 ^ hi) << 6)), xn = xc;
        bf16x8 wc = *reinterpret_cast<const bf16x8*>(Wn), wn = wc;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (ks + 1 < KS) {
            xn = *reinterpret_cast<const bf16x8*>(A + 1024 + (((ks + 1) ^ hi) << 6));
            wn = *reinterpret_cast<const bf16x8*>(Wn + (ks + 1) * 64);
          }
          acc[0] = mma16(wi[0][ks], xc, acc[0]);
          acc[1] = mma16(wi[1][ks], xc, acc[1]);
          acc[3] = mma16(wc, xc, acc[3]);
          xc = xn; wc = wn;
        }
      }
      // ---- epilogue on 4 units of sequence b
      float r[4], z[4], n[4], hn[4], hp[4], h[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hp[e] = hp4[e];
        r[e] = sigmoidf_(gi[0][e] + gz[0][e] + acc[0][e]);
        z[e] = sigmoidf_(gi[1][e] + gz[1][e] + acc[1][e]);
        hn[e] = acc[2][e] + bh[e];
        n[e] = tanhf_(gi[2][e] + gz[2][e] + acc[3][e] + r[e] * hn[e]);
        h[e] = (1.0f - z[e]) * n[e] + z[e] * hp[e];
      }
      // the next tile must have landed before the barrier; waiting HERE (before this tile's stores are issued) keeps the
      // write-through stores out of the wait - they drain under the next tile's MFMAs
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (live) {
        *reinterpret_cast<f32x4*>(carry + (long)b * H + u0) = f32x4{h[0], h[1], h[2], h[3]};
        const long s = (long)(o + b) * H + u0;
        store_sc1_b64(r_hseq, (unsigned)(s * 2), to_bf16x4(h));
        if (hdrop) {
          float hd[4], m[4];
          dropout_scale4(a.seed, l, t, b * a.b_mul + a.b_add, u0 >> 2, a.p_drop, inv_keep, m);
#pragma unroll
          for (int e = 0; e < 4; ++e) hd[e] = h[e] * m[e];
          store_sc1_b64(r_hdrop, (unsigned)(s * 2), to_bf16x4(hd));
        }
        if (a.sv[l][0]) {
          *reinterpret_cast<bf16x4*>(a.sv[l][0] + s) = to_bf16x4(r);
          *reinterpret_cast<bf16x4*>(a.sv[l][1] + s) = to_bf16x4(z);
          *reinterpret_cast<bf16x4*>(a.sv[l][2] + s) = to_bf16x4(n);
          *reinterpret_cast<bf16x4*>(a.sv[l][3] + s) = to_bf16x4(hn);
          *reinterpret_cast<bf16x4*>(a.sv[l][4] + s) = to_bf16x4(hp);
        }
      }
      if (j == 0) PG_STAMP(4);
    }
    PG_STAMP(5);
    // ---- publish step t of this unit slice
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PG_STAMP(6);
    if (tid == 0) __hip_atomic_fetch_add(cnt_own, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PG_STAMP(7);
  }
}

template <int H>
__global__ __launch_bounds__(256) void dec_fwd_persist_kernel(DecFwdPersist a, Sched sd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int g = blockIdx.x % a.NG, k = blockIdx.x / a.NG;   // blockIdx % 8 labels the XCD: a group's workgroups share one
  const int l = k / DecGeo<H>::NCU, c = k % DecGeo<H>::NCU;
  if (l == 0) dec_fwd_body<H, true>(a, sd, g, l, c, smem);
  else dec_fwd_body<H, false>(a, sd, g, l, c, smem);
}

unsigned long long* g_stamps = nullptr;
int* g_fail_host = nullptr;        // pinned mirror of the sticky give-up word
unsigned* g_fail_dev = nullptr;   // never cleared: one timed-out launch condemns the process

}  // namespace

bool dec_persist_supported(int H, int L, int B) { return H == 512 && L >= 1 && L * (H / 64) * 8 <= 256 && B >= 1; }

int persist_poll_failure() { return g_fail_host ? *reinterpret_cast<volatile int*>(g_fail_host) : 0; }

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

int persist_fail_mirror(hipStream_t stream) {
  PG_CHECK_HIP(hipMemcpyAsync(g_fail_host, g_fail_dev, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
  return PG_OK;
}

int launch_dec_fwd_persist(DecFwdPersist& a, const Sched& sd, int H, hipStream_t stream) {
  PG_REQUIRE(dec_persist_supported(H, a.L, a.B), "dec_fwd_persist: shape H=%d L=%d not covered", H, a.L);
  PG_REQUIRE(a.sync && a.table && a.tok && a.zproj && a.h0_op && a.h0_32, "dec_fwd_persist: NULL buffer");
  using G = DecGeo<512>;
  a.NG = 8;   // one group per XCD
  PG_TRY(persist_fail_words(&a.fail));
  a.stamps = g_stamps;
  PG_CHECK_HIP(hipMemsetAsync(a.sync, 0, PG_PERSIST_SYNC_WORDS * sizeof(unsigned), stream));
  static bool attr_done = false;
  if (!attr_done) {
    PG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fwd_persist_kernel<512>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM));
    attr_done = true;
  }
  // Every workgroup waits for its peers, so the whole grid must be co-resident: one workgroup per CU (512 registers per
  // lane, 130 KB LDS) on a device with at least `grid` CUs.  Checked here; a plain launch has the same residency as a
  // cooperative one and works when several processes use the GPU (their kernels drain independently, waits are bounded).
  const int grid = a.NG * a.L * G::NCU;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    PG_CHECK_HIP(hipGetDevice(&dev));
    PG_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  PG_REQUIRE(grid <= n_cu, "dec_fwd_persist: %d workgroups cannot be co-resident on %d CUs", grid, n_cu);
  hipLaunchKernelGGL((dec_fwd_persist_kernel<512>), dim3(grid), dim3(256), (size_t)G::SMEM, stream, a, sd);
  PG_CHECK_LAUNCH();
  PG_TRY(persist_fail_mirror(stream));
  return PG_OK;
}

// development aid (tools/persist_stamps.py): device buffer of 8 + 64 u64 words; word 0 = layer to trace
extern "C" __attribute__((visibility("default"))) int pg_persist_stamps(void* buf) {
  g_stamps = static_cast<unsigned long long*>(buf);
  return PG_OK;
}
