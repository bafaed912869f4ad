// Row-resident persistent encoder recurrence (see pg_persist.h).  bf16 operands, fp32 state and accumulation.
//
// Geometry (H = 256): block = 16 sequences x all H units, 8 waves; wave w owns units [32w, 32w+32) as two 16-unit MFMA
// tiles.  MFMA v_mfma_f32_16x16x32_bf16 is issued with the WEIGHTS as its first operand and the activations as its second,
// so the accumulator tile is [unit][sequence]: lane (lr = lane & 15, lq = lane >> 4) holds units 4lq..4lq+3 of sequence lr.
// Four consecutive units per lane make every global access of the epilogue an 8-byte (bf16x4) or 16-byte (float4) one.
#include <hip/hip_runtime.h>
#include "pg_mma.h"
#include "pg_persist.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

namespace {

constexpr int PR = 16;   // sequences per block

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. for the acknowledgement
// of the step's global stores (2-3 us each on this chip) and for the prefetched loads of the next step - per step of a
// ~1 us recurrence.  What the waves exchange at these barriers is the LDS tile, nothing else.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// 8-byte store through a buffer descriptor: lanes whose sequence is over get an offset beyond the descriptor's range and
// the hardware drops them, so every wave issues the same number of stores and hipcc's vmcnt bookkeeping stays exact
__device__ __forceinline__ void store_b64(__amdgpu_buffer_rsrc_t r, bool ok, long elem, const bf16x4& v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)(ok ? (unsigned)(elem * 2) : 0x80000000u), 0, 0);
}

__device__ __forceinline__ f32x4 mma16(const bf16x8& w, const bf16x8& x, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 to_bf16x4(const float (&v)[4]) {
  bf16x4 o;
  o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
  return o;
}

template <int H> struct PersistGeo {
  static constexpr int NW = H / 32;              // waves
  static constexpr int NTH = 64 * NW;
  static constexpr int KS_H = H / 32;            // MFMA K steps over a K = H contraction
  static constexpr int ROWB_H = H * 2 + 16;      // forward kernel: padded LDS row of H bf16
  static constexpr int ROWB_X = H * 2;           // backward kernel: unpadded, 16-byte chunk q of row r at position q ^ (r & 15)
  static constexpr int ROWB_3H = 3 * H * 2;      // the same layout (XOR inside each aligned group of 16 chunks)
  static constexpr int WN_BYTES = H * ROWB_H;    // n-gate weight rows kept in LDS
  static constexpr int FWD_SMEM = WN_BYTES + 2 * PR * ROWB_H;
  static constexpr int BWD_SMEM = H * ROWB_X + PR * ROWB_3H + PR * ROWB_X;   // + the dn tile (the n gate of dGI)
};

// first element of the lane-major slot of (step t, block at row0): packed offset of the step + 16 rows of padding per
// step, so that the 16-row slot of a step's last, partial block never reaches the next step's slots
__device__ __forceinline__ long sv_slot(int off_t, int t, int row0, int H) { return (long)(off_t + 16 * t + row0) * H; }

// steps this block runs = length of its longest (first) sequence = #t with B_t > row0 (B_t is non-increasing)
__device__ __forceinline__ int block_steps(const Sched& sd, int row0) {
  int n = 0;
  while (n < sd.T && sd.off[n + 1] - sd.off[n] > row0) ++n;
  return n;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(PersistGeo<H>::NTH) void enc_fwd_persist_kernel(EncFwdPersist a, Sched sd) {
  using G = PersistGeo<H>;
  constexpr int KS = G::KS_H, ROWB = G::ROWB_H;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sWn = smem;                 // [H][ROWB]   W_hn rows
  char* sA = smem + G::WN_BYTES;    // [2][PR][ROWB] bf16 state tile, double buffered
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int row0 = blockIdx.x * PR;
  const int b = row0 + lr;                       // this lane's sequence
  const int Tb = block_steps(sd, row0);
  const int u0 = wave * 32 + lq * 4;             // first unit of tile 0 held by this lane (tile 1: +16)

  // ---- stationary weights: r and z rows of this wave's units -> registers, every n row -> LDS
  bf16x8 wreg[2][2][KS];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wreg[g][nt][ks] = *reinterpret_cast<const bf16x8*>(a.Whh + (long)(g * H + wave * 32 + nt * 16 + lr) * H + (ks * 4 + lq) * 8);
  for (int c = tid; c < H * (H / 8); c += G::NTH) {
    const int u = c / (H / 8), ch = c % (H / 8);
    *reinterpret_cast<uint4*>(sWn + u * ROWB + ch * 16) = *reinterpret_cast<const uint4*>(a.Whh + (long)(2 * H + u) * H + ch * 8);
  }
  for (int c = tid; c < 2 * PR * ROWB / 16; c += G::NTH) reinterpret_cast<uint4*>(sA)[c] = make_uint4(0, 0, 0, 0);
  float bh[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) bh[nt][i] = a.bhn[u0 + nt * 16 + i];

  float h[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) h[nt][i] = 0.0f;

  // token of this lane's sequence at step t (0 when the sequence is over: any valid table row, the value is unused)
  auto token_at = [&](int t) __attribute__((always_inline)) -> int {
    if (t >= Tb) return 0;
    const int o = sd.off[t], Bt = sd.off[t + 1] - o;
    return b < Bt ? a.tok[o + b] : 0;
  };
  auto gather = [&](int tok, f32x4 (&gi)[3][2]) __attribute__((always_inline)) {
    const float* p = a.table + (long)tok * 3 * H + u0;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) gi[g][nt] = *reinterpret_cast<const f32x4*>(p + g * H + nt * 16);
  };
  f32x4 gi[3][2];
  gather(token_at(0), gi);
  int tok_next = token_at(1);
  __syncthreads();

  for (int t = 0; t < Tb; ++t) {
    const int cur = t & 1;
    // prefetch: gather-table rows of step t+1, token of step t+2
    f32x4 gin[3][2];
    gather(tok_next, gin);
    tok_next = token_at(t + 2);
    // gates: acc[g][nt][unit 4lq+i][sequence lr] = sum_k W[g*H + unit][k] * h_{t-1}[sequence][k]
    f32x4 acc[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[g][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* A = sA + cur * (PR * ROWB) + lr * ROWB + lq * 16;
    const char* Wn = sWn + (wave * 32 + lr) * ROWB + lq * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(A + ks * 64);
      const bf16x8 n0 = *reinterpret_cast<const bf16x8*>(Wn + ks * 64);
      const bf16x8 n1 = *reinterpret_cast<const bf16x8*>(Wn + 16 * ROWB + ks * 64);
      acc[0][0] = mma16(wreg[0][0][ks], x, acc[0][0]);
      acc[0][1] = mma16(wreg[0][1][ks], x, acc[0][1]);
      acc[1][0] = mma16(wreg[1][0][ks], x, acc[1][0]);
      acc[1][1] = mma16(wreg[1][1][ks], x, acc[1][1]);
      acc[2][0] = mma16(n0, x, acc[2][0]);
      acc[2][1] = mma16(n1, x, acc[2][1]);
    }
    // epilogue on this lane's 2 x 4 units of sequence b
    const int o = sd.off[t], Bt = sd.off[t + 1] - o;
    const bool live = b < Bt;
    char* An = sA + (cur ^ 1) * (PR * ROWB) + lr * ROWB;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float r[4], z[4], n[4], hn[4], hp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hp[i] = h[nt][i];
        r[i] = sigmoidf_(gi[0][nt][i] + acc[0][nt][i]);
        z[i] = sigmoidf_(gi[1][nt][i] + acc[1][nt][i]);
        hn[i] = acc[2][nt][i] + bh[nt][i];
        n[i] = tanhf_(gi[2][nt][i] + r[i] * hn[i]);
        if (live) h[nt][i] = (1.0f - z[i]) * n[i] + z[i] * hp[i];
      }
      const int u = u0 + nt * 16;
      if (a.sv_r) {   // kernel-uniform
        // r, z, n, hn are read back by enc_bwd_persist_kernel only, lane for lane: they are stored in LANE order (512
        // contiguous bytes per wave instruction) in the 16-row slot of (step, block).  The row-major form of the same
        // data is 16 rows x 32 bytes per instruction, which the address path serialises (16 cache lines): 10 such
        // stores per wave and step were most of the step's time.  h_prev stays row-major: it is a GEMM operand.
        const long sl = sv_slot(o, t, row0, H) + ((wave * 2 + nt) * 64 + lane) * 4;
        *reinterpret_cast<bf16x4*>(a.sv_r + sl) = to_bf16x4(r);
        *reinterpret_cast<bf16x4*>(a.sv_z + sl) = to_bf16x4(z);
        *reinterpret_cast<bf16x4*>(a.sv_n + sl) = to_bf16x4(n);
        *reinterpret_cast<bf16x4*>(a.sv_hn + sl) = to_bf16x4(hn);
        if (live) *reinterpret_cast<bf16x4*>(a.sv_hprev + (long)(o + b) * H + u) = to_bf16x4(hp);
      }
      *reinterpret_cast<bf16x4*>(An + u * 2) = to_bf16x4(h[nt]);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) gi[g][nt] = gin[g][nt];
    __syncthreads();
  }
  if (b < a.B) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const long s = (long)b * H + u0 + nt * 16;
      *reinterpret_cast<f32x4*>(a.h_fin32 + s) = f32x4{h[nt][0], h[nt][1], h[nt][2], h[nt][3]};
      *reinterpret_cast<bf16x4*>(a.h_fin_op + s) = to_bf16x4(h[nt]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward: dh_{t}[b, j] = z.dh_{t+1} (carried in registers) + sum_k dGH_{t+1}[b, k] W_hh[k, j]   (k over the 3H gate
// columns), then the gate gradients of step t from the saved activations.  The wave owns output units j in [32w, 32w+32);
// W_hh^T rows of those units: k in [0, 2H) (r, z columns) in registers, k in [2H, 3H) (n columns) of every unit in LDS.
// ------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(PersistGeo<H>::NTH) void enc_bwd_persist_kernel(EncBwdPersist a, Sched sd) {
  using G = PersistGeo<H>;
  constexpr int KS = G::KS_H, ROWB = G::ROWB_X, ROWA = G::ROWB_3H;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sWn = smem;                 // [H][ROWB]: W_hh^T[j][2H + k]
  char* sA = smem + H * ROWB;       // [PR][ROWA]: dGH_{t+1} of the block's sequences (bf16)
  char* sDn = sA + PR * ROWA;       // [PR][ROWB]: dn_t, the one gate in which dGI differs from dGH
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int row0 = blockIdx.x * PR;
  const int b = row0 + lr;
  const int Tb = block_steps(sd, row0);
  const int u0 = wave * 32 + lq * 4;

  bf16x8 wreg[2][2 * KS];   // [unit tile][k step over the r|z columns]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int ks = 0; ks < 2 * KS; ++ks)
      wreg[nt][ks] = *reinterpret_cast<const bf16x8*>(a.WhhT + (long)(wave * 32 + nt * 16 + lr) * 3 * H + (ks * 4 + lq) * 8);
  for (int c = tid; c < H * (H / 8); c += G::NTH) {
    const int u = c / (H / 8), ch = c % (H / 8);
    *reinterpret_cast<uint4*>(sWn + u * ROWB + ((ch ^ (u & 15)) << 4)) = *reinterpret_cast<const uint4*>(a.WhhT + (long)u * 3 * H + 2 * H + ch * 8);
  }
  for (int c = tid; c < PR * ROWA / 16; c += G::NTH) reinterpret_cast<uint4*>(sA)[c] = make_uint4(0, 0, 0, 0);

  struct Saved { bf16x4 r[2], z[2], n[2], hn[2], hp[2]; };
  auto load_saved = [&](int t, Saved& s) __attribute__((always_inline)) {
    // sequences that are over at step t read row 0 of the step (valid memory, values unused)
    const int o = t >= 0 ? sd.off[t] : 0, Bt = t >= 0 ? sd.off[t + 1] - o : 0;
    const long base = (long)(o + (b < Bt ? b : 0)) * H + u0;
    const int tt = t >= 0 ? t : 0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const long sl = sv_slot(o, tt, row0, H) + ((wave * 2 + nt) * 64 + lane) * 4;   // lane order, as the forward kernel wrote it
      s.r[nt] = *reinterpret_cast<const bf16x4*>(a.sv_r + sl);
      s.z[nt] = *reinterpret_cast<const bf16x4*>(a.sv_z + sl);
      s.n[nt] = *reinterpret_cast<const bf16x4*>(a.sv_n + sl);
      s.hn[nt] = *reinterpret_cast<const bf16x4*>(a.sv_hn + sl);
      s.hp[nt] = *reinterpret_cast<const bf16x4*>(a.sv_hprev + base + nt * 16);
    }
  };
  float dhc[2][4];   // direct path z.dh carried from step t+1
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) dhc[nt][i] = 0.0f;
  Saved sv, sv1;   // saved activations of step t and (in flight) of step t-1
  if (Tb > 0) load_saved(Tb - 1, sv);
  const __amdgpu_buffer_rsrc_t r_dGI = pg_make_rsrc(a.dGI), r_dGH = pg_make_rsrc(a.dGH);
  // gradient injected at this lane's sequence's last step: loaded once.  A load inside the step loop - even a
  // conditional one - makes hipcc wait for vmcnt(0) at the join, i.e. for the previous step's store acknowledgements and
  // the prefetched activations, every step.
  f32x4 inj[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
    inj[nt] = b < a.B ? *reinterpret_cast<const f32x4*>(a.dhfin + (long)b * H + u0 + nt * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // one step; `sv` holds the saved activations of step t (requested a step ago), `svn` receives those of step t-1.
  // The two register sets swap by NAME (the loop below is unrolled by two): a register copy of a set still in flight
  // would make hipcc wait for it - and, vmcnt being one in-order counter, for every store issued before it.
  auto step = [&](int t, const Saved& sv, Saved& svn) __attribute__((always_inline)) {
    load_saved(t - 1, svn);   // t - 1 < 0 reads step 0's first row, unused
    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t + 1 < Tb) {   // block-uniform: the first step has no recurrent gradient yet
      const int hi = lr >> 2;   // chunk q of row lr sits at q ^ lr (see the forward kernel)
      const char* A = sA + lr * ROWA + ((lq ^ lr) & 3) * 16;
      const char* Wn = sWn + (wave * 32 + lr) * ROWB + ((lq ^ lr) & 3) * 16;
#pragma unroll
      for (int ks = 0; ks < 2 * KS; ++ks) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(A + ((ks ^ hi) << 6));
        acc[0] = mma16(wreg[0][ks], x, acc[0]);
        acc[1] = mma16(wreg[1][ks], x, acc[1]);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(A + 4 * H + ((ks ^ hi) << 6));
        const bf16x8 n0 = *reinterpret_cast<const bf16x8*>(Wn + ((ks ^ hi) << 6));
        const bf16x8 n1 = *reinterpret_cast<const bf16x8*>(Wn + 16 * ROWB + ((ks ^ hi) << 6));
        acc[0] = mma16(n0, x, acc[0]);
        acc[1] = mma16(n1, x, acc[1]);
      }
    }
    const int o = sd.off[t], Bt = sd.off[t + 1] - o;
    const int Bt1 = (t + 1 < sd.T) ? sd.off[t + 2] - sd.off[t + 1] : 0;
    const bool live = b < Bt, first = live && b >= Bt1;   // first: step t is this sequence's last step
    float dr[2][4], dzg[2][4], dn[2][4], dnr[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float dh[4];
      if (first) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dh[i] = inj[nt][i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) dh[i] = dhc[nt][i] + acc[nt][i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float r = (float)sv.r[nt][i], z = (float)sv.z[nt][i], n = (float)sv.n[nt][i];
        const float hn = (float)sv.hn[nt][i], hp = (float)sv.hp[nt][i];
        const float d = live ? dh[i] : 0.0f;
        dn[nt][i] = d * (1.0f - z) * (1.0f - n * n);
        dzg[nt][i] = d * (hp - n) * z * (1.0f - z);
        dr[nt][i] = dn[nt][i] * hn * r * (1.0f - r);
        dnr[nt][i] = dn[nt][i] * r;
        dhc[nt][i] = d * z;
      }
    }
    lds_barrier();   // every wave has finished reading dGH_{t+1}
    {
      char* An = sA + lr * ROWA;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int u = u0 + nt * 16;
        const int sw = (((u >> 3) ^ lr) << 4) + (u & 7) * 2;   // H * 2 bytes = 32 chunks: the gate offset keeps the XOR group
        *reinterpret_cast<bf16x4*>(An + sw) = to_bf16x4(dr[nt]);
        *reinterpret_cast<bf16x4*>(An + H * 2 + sw) = to_bf16x4(dzg[nt]);
        *reinterpret_cast<bf16x4*>(An + 2 * H * 2 + sw) = to_bf16x4(dnr[nt]);
        *reinterpret_cast<bf16x4*>(sDn + lr * ROWB + sw) = to_bf16x4(dn[nt]);
      }
    }
    lds_barrier();
    // dGH_t and dGI_t go to memory from the LDS tile, row-major and coalesced: 16-byte chunk c of row r sits at position
    // c ^ (r & 15) of its 16-chunk group.  (As 12 per-lane stores of 16 rows x 32 bytes they cost the step more than its
    // arithmetic.)  The tile is next written after the next step's first barrier, which every wave reaches after these reads.
    {
      const unsigned oob = 0x80000000u;
      constexpr int CPR = 3 * H / 8;               // 16-byte chunks per row of a [., 3H] bf16 array
#pragma unroll
      for (int k = 0; k < PR * CPR / G::NTH; ++k) {
        const int idx = tid + k * G::NTH;
        const int row = idx / CPR, c = idx % CPR;
        const bool ok = row0 + row < Bt;
        const unsigned goff = ok ? (unsigned)(((long)(o + row0 + row) * 3 * H) * 2 + c * 16) : oob;
        const u32x4 vh = *reinterpret_cast<const u32x4*>(sA + row * ROWA + ((c ^ (row & 15)) << 4));
        __builtin_amdgcn_raw_buffer_store_b128(vh, r_dGH, (int)goff, 0, 0);
        const u32x4 vi = c < 2 * H / 8 ? vh : *reinterpret_cast<const u32x4*>(sDn + row * ROWB + (((c - 2 * H / 8) ^ (row & 15)) << 4));
        __builtin_amdgcn_raw_buffer_store_b128(vi, r_dGI, (int)goff, 0, 0);
      }
    }
  };
  for (int t = Tb - 1; t >= 0; t -= 2) {
    step(t, sv, sv1);
    if (t - 1 >= 0) step(t - 1, sv1, sv);
  }
}

}  // namespace

bool enc_persist_supported(int H) { return H == 256; }

int launch_enc_fwd_persist(const EncFwdPersist& a, const Sched& sd, int H, hipStream_t stream) {
  PG_REQUIRE(enc_persist_supported(H), "enc_fwd_persist: hidden size %d not covered", H);
  PG_REQUIRE(a.Whh && a.table && a.bhn && a.tok && a.h_fin32 && a.h_fin_op && a.B > 0, "enc_fwd_persist: NULL buffer");
  using G = PersistGeo<256>;
  PG_LAUNCH_BIG_SMEM((enc_fwd_persist_kernel<256>), dim3(cdiv(a.B, PR)), dim3(G::NTH), (size_t)G::FWD_SMEM, stream, a, sd);
  PG_CHECK_LAUNCH();
  return PG_OK;
}

int launch_enc_bwd_persist(const EncBwdPersist& a, const Sched& sd, int H, hipStream_t stream) {
  PG_REQUIRE(enc_persist_supported(H), "enc_bwd_persist: hidden size %d not covered", H);
  PG_REQUIRE(a.WhhT && a.dhfin && a.sv_r && a.dGI && a.dGH && a.B > 0, "enc_bwd_persist: NULL buffer");
  using G = PersistGeo<256>;
  PG_LAUNCH_BIG_SMEM((enc_bwd_persist_kernel<256>), dim3(cdiv(a.B, PR)), dim3(G::NTH), (size_t)G::BWD_SMEM, stream, a, sd);
  PG_CHECK_LAUNCH();
  return PG_OK;
}
