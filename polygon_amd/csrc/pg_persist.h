// Row-resident persistent recurrence kernels (bf16): ONE launch runs every time step of the encoder GRU.
//
// The recurrence of a GRU is independent per sequence, so a workgroup that owns 16 sequences can walk all their time
// steps without ever talking to another workgroup - provided it can hold the recurrent weight matrix.  On gfx950 it
// can: W_hh of the encoder is 3H x H bf16 = 384 KiB at H = 256, and one CU has 512 KiB of vector registers plus
// 160 KiB of LDS.  Each of the 8 waves keeps the r and z gate rows of its 32 hidden units in registers (128 VGPRs of
// MFMA fragments); the n gate rows of all waves live in LDS (132 KiB); the 16 x H hidden-state tile bounces between
// two small LDS buffers.  Per step a block issues 48 MFMAs per wave and touches global memory only for the token's
// gather-table row and the activations saved for BPTT - no launch, no weight traffic, no inter-block hand-off.
// Replaces T launches of the step kernel (each ~11 us, latency-bound) by one launch of ~T x 1 us.
#pragma once
#include "pg_common.h"
#include "pg_layout.h"

#ifndef PG_TRY
#define PG_TRY(expr)            \
  do {                          \
    int _s = (expr);            \
    if (_s != PG_OK) return _s; \
  } while (0)
#endif

struct EncFwdPersist {
  const bf16_t* Whh = nullptr;     // [3H, H] (gate-major rows r|z|n, K contiguous)
  const float* table = nullptr;    // [V, 3H]: x_emb[v].W_ih^T + b_ih + (b_hr | b_hz | 0)
  const float* bhn = nullptr;      // [H]
  const int* tok = nullptr;        // packed time-major tokens [N]
  bf16_t *sv_r = nullptr, *sv_z = nullptr, *sv_n = nullptr, *sv_hn = nullptr, *sv_hprev = nullptr;   // packed [N, H] or null
  float* h_fin32 = nullptr;        // [B, H] each row's state after its own last step
  bf16_t* h_fin_op = nullptr;      // [B, H] the same in bf16
  int B = 0;
};

struct EncBwdPersist {
  const bf16_t* WhhT = nullptr;    // [H, 3H] (W_hh transposed, K = gate column contiguous)
  const float* dhfin = nullptr;    // [B, H] gradient w.r.t. each row's final state
  const bf16_t *sv_r = nullptr, *sv_z = nullptr, *sv_n = nullptr, *sv_hn = nullptr, *sv_hprev = nullptr;   // packed [N, H]
  bf16_t *dGI = nullptr, *dGH = nullptr;   // packed [N, 3H]
  int B = 0;
};

// Decoder stack (L layers, H = 512), forward with teacher forcing.  The weights of one layer no longer fit one CU, so a
// GROUP of L * H/64 workgroups shares them: workgroup (layer l, slice c) keeps the W_hh / W_ih rows of 64 hidden units
// stationary in its registers (4 waves x 16 units x 3 gates x K = 2H) and every group walks the time steps of its own
// 1/NG of the batch (sequences g, g+NG, g+2NG, ... so that all groups shrink together as sequences end).  Per step a
// workgroup needs the whole h_{l,t-1} and x = h_{l-1,t} rows of its sequences: it reads them from the hseq arrays the
// other workgroups of the group publish (write-through stores, then one arrival on a per-(group, layer) counter) - an
// all-gather through the memory side of the L2s, never a grid-wide barrier: layer l runs step t as soon as (l, t-1)
// and (l-1, t) are complete, so the layers pipeline by themselves.
struct DecFwdPersist {
  int L = 0, B = 0, NG = 0;
  const bf16_t* Whh[PG_MAX_LAYERS] = {};      // [3H, H]
  const bf16_t* Wih[PG_MAX_LAYERS] = {};      // [3H, H] for l >= 1 (layer 0 takes its input through the gather table)
  const float* bhn[PG_MAX_LAYERS] = {};       // [H]
  const float* bias_gi[PG_MAX_LAYERS] = {};   // [3H] for l >= 1: b_ih + (b_hr | b_hz | 0)
  const float* table = nullptr;               // layer 0: [V, 3H]
  const int* tok = nullptr;                   // packed decoder input tokens [N]
  const float* zproj = nullptr;               // [B, 3H]
  const bf16_t* h0_op = nullptr;              // [B, H]
  const float* h0_32 = nullptr;               // [B, H]
  bf16_t* hseq[PG_MAX_LAYERS] = {};           // packed [N, H]: layer outputs (published between workgroups)
  bf16_t* hdrop[PG_MAX_LAYERS] = {};          // packed [N, H]: dropout-masked outputs (null: no dropout / top layer)
  float* carry[PG_MAX_LAYERS] = {};           // [B, H] fp32 state
  bf16_t* sv[PG_MAX_LAYERS][5] = {};          // packed [N, H] r, z, n, hn, hprev (null: inference)
  int sv_lane = 0;                            // sv[l][0..3] in lane order (sv_tile_slot) instead of packed rows: see pg_persist_dec.hip
  float p_drop = 0.0f;
  uint64_t seed = 0;
  int b_mul = 1, b_add = 0;
  unsigned* sync = nullptr;                   // PG_PERSIST_SYNC_WORDS words, zeroed by the launcher
  unsigned* fail = nullptr;                   // set by the launcher: process-wide sticky give-up word
  unsigned spin_limit = 0;                    // set by the launcher: polls before a waiting workgroup gives up
  unsigned long long* stamps = nullptr;       // development aid: wall-clock stamps of one workgroup (pg_persist_stamps)
  int dbg = 0;                                // development aid (PG_PERSIST_DBG): 1 no DMA, 2 no MFMA, 4 no epilogue stores, 8 no epilogue loads
};
constexpr int PG_PERSIST_MAX_GROUPS = 16;
constexpr int PG_PERSIST_SYNC_WORDS = PG_PERSIST_MAX_GROUPS * PG_MAX_LAYERS * 16 + 16;   // counters [group][layer] at a 16-word stride, then the give-up word

// Number of groups for a batch: 8 (one per XCD, 192 workgroups) or - with 3 layers, when it saves a 16-row tile per step
// - 10: groups 8 and 9 take the 8 spare CUs of XCDs 0-2 and 3-5, one LAYER per XCD, so that the 8 slices of a layer still
// share an L2 (256 workgroups, the 16 on XCDs 6-7 idle).  Placement is a speed matter only.
static inline int persist_groups(int L, int B) {
  auto tiles = [&](int ng) { return ((B + ng - 1) / ng + 15) / 16; };
  static const int forced = getenv("PG_PERSIST_GROUPS") ? atoi(getenv("PG_PERSIST_GROUPS")) : 0;   // A/B aid: 8 or 10
  if (forced == 8 || (forced == 10 && L == 3)) return forced;
  return (L == 3 && tiles(10) < tiles(8)) ? 10 : 8;
}
// workgroup -> (group, layer-order index k/NCU, slice); false: idle workgroup
__device__ __forceinline__ bool persist_map(int bid, int NG, int NCU, int& g, int& li, int& c) {
  if (NG == 8) {
    g = bid % 8;
    const int k = bid / 8;
    li = k / NCU; c = k % NCU;
    return true;
  }
  const int xcd = bid % 8, slot = bid / 8;           // NG == 10: 32 slots per XCD, NCU == 8, 3 layers
  if (slot < 3 * NCU) { g = xcd; li = slot / NCU; c = slot % NCU; return true; }
  if (xcd >= 6) return false;
  g = 8 + xcd / 3; li = xcd % 3; c = slot - 3 * NCU;
  return true;
}

// Lane-order storage of the saved gate activations r, z, n, hn (read back by the BPTT kernel only, lane for lane): the
// 16-row tile j of group g at step t owns slot sv_tile_slot(...) of 16 x H elements, laid out [slice 8][wave 4][lane 64][4].
// Tiles of step t start at floor(off[t] / 16) + (NG + 1) * t, which leaves room for every group's partial last tile:
// at most N / 16 + (NG + 1) * T + 1 slots.
// The two arrival counters a step waits for, polled AT ONCE: lane 0 the first, lane 1 the second.  One after the other the
// second poll costs an L2 round trip of its own (0.4-0.6 us under load) even when its counter has long been there.  Called
// by a whole wave; the result is wave-uniform.  Bounded like the single poll (pg_persist_dec.hip, protocol): the lane that
// runs out of polls sets the launch's error word and the sticky give-up word.
__device__ __forceinline__ bool pg_wait_ge2(unsigned* p0, unsigned t0, unsigned* p1, unsigned t1, bool two, unsigned* err,
                                            unsigned* sticky, unsigned limit) {
  const int lane = threadIdx.x & 63;
  unsigned* p = lane == 0 ? p0 : p1;
  const unsigned target = lane == 0 ? t0 : t1;
  bool pending = lane == 0 || (two && lane == 1);
  for (unsigned spins = 0;; ++spins) {
    if (pending) pending = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target;
    if (__ballot(pending) == 0ull) return true;
    if ((spins & 63u) == 63u) {
      const unsigned e = lane == 0 ? __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      if (__ballot(e != 0u) != 0ull) return false;
    }
    if (spins > limit) {
      if (lane == 0) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

__device__ __forceinline__ long sv_tile_slot(int off_t, int t, int NG, int j, int g) {
  return (long)(off_t >> 4) + (long)(NG + 1) * t + (long)j * NG + g;
}
constexpr long PG_SV_LANE_PAD_ROWS = 16L * (10 + 1) * 512 + 16;   // rows beyond the packed count (NG <= 10, T <= 512)

// Decoder stack, backward (BPTT): the same grouping, workgroup (layer l, slice c) keeps the rows of W_hh_l^T and
// W_ih_{l+1}^T that produce dh of its 64 units.  It consumes dGH_{l,t+1} (own layer) and dGI_{l+1,t} (layer above) as
// published by the other workgroups and publishes dGI_{l,t} / dGH_{l,t}; see pg_persist_bwd.hip.
struct DecBwdPersist {
  int L = 0, B = 0, NG = 0;
  const bf16_t* WhhT[PG_MAX_LAYERS] = {};     // [H, 3H]
  const bf16_t* WihT[PG_MAX_LAYERS] = {};     // [H, 3H] of layer l >= 1 (read by the workgroups of layer l-1)
  const float* dout = nullptr;                // packed [N, H]: gradient of the top layer's outputs (vocab projection)
  const bf16_t* sv[PG_MAX_LAYERS][5] = {};    // packed [N, H] r, z, n, hn, hprev saved by the forward pass
  bf16_t* dGI[PG_MAX_LAYERS] = {};            // packed [N, 3H] (published)
  bf16_t* dGH[PG_MAX_LAYERS] = {};            // packed [N, 3H] (published)
  float* carry[PG_MAX_LAYERS] = {};           // [B, H] z.dh carried to the previous step
  float* dh0[PG_MAX_LAYERS] = {};             // [B, H] per-layer gradient of h0 (summed by the launcher)
  // bias gradients (nullable, fp32 [3H], accumulated atomically at the end of the launch): d b_hh[l] = column sums of dGH_l,
  // d b_ih[l] = column sums of dGI_l - every lane keeps the running sums of the gate gradients it produces anyway, which
  // takes these sums out of the weight-gradient GEMMs (their fused column sums cost the 256 x 256 kernel 60 % of its time)
  float* dbhh[PG_MAX_LAYERS] = {};
  float* dbih[PG_MAX_LAYERS] = {};
  int sv_lane = 0;                            // sv[l][0..3] are in lane order, as the forward kernel wrote them
  float p_drop = 0.0f;
  uint64_t seed = 0;
  int b_mul = 1, b_add = 0;
  unsigned* sync = nullptr;                   // PG_PERSIST_SYNC_WORDS words, zeroed by the launcher
  unsigned* fail = nullptr;                   // set by the launcher
  unsigned spin_limit = 0;                    // set by the launcher
  unsigned long long* stamps = nullptr;       // development aid (pg_persist_stamps)
  int dbg = 0;                                // development aid (PG_PERSIST_DBG): 1 no DMA, 2 no MFMA, 4 no epilogue I/O
};

// Autoregressive sampler, whole decode loop in one launch (pg_persist_sample.hip): 8 groups x (3 layers x 8 slices + 8
// head workgroups); a group's rows travel in S sub-batches of TS 16-row tiles round the stages l0 -> l1 -> l2 -> head.
constexpr int PG_SAMPLE_MAX_SUB = 8;          // sub-batches per group
constexpr int PG_SAMPLE_MAX_TS = 16;          // 16-row tiles per sub-batch (the token block of layer 0 in LDS)
constexpr int PG_SAMPLE_SYNC_WORDS = 8 * 4 * PG_SAMPLE_MAX_SUB * 16 + 16;   // counters [group][stage][sub-batch] at a 16-word stride, then the give-up word
struct DecSamplePersist {
  int L = 0, n = 0, T = 0;                    // rows, decode steps (max_len - 1)
  int ntile = 0, S = 0, TS = 0;               // set by the launcher
  const bf16_t* Whh[PG_MAX_LAYERS] = {};      // [3H, H]
  const bf16_t* Wih[PG_MAX_LAYERS] = {};      // [3H, H] for l >= 1
  const float* bhn[PG_MAX_LAYERS] = {};       // [H]
  const float* bias_gi[PG_MAX_LAYERS] = {};   // [3H] for l >= 1
  const float* table = nullptr;               // layer 0: [V, 3H]
  const float* zproj = nullptr;               // [n, 3H]
  bf16_t* hop[PG_MAX_LAYERS][2] = {};         // [n, H] layer outputs, ping-pong by step parity; [l][0] = h0 on entry
  float* carry[PG_MAX_LAYERS] = {};           // [n, H] fp32 state; = h0 on entry
  int* tok = nullptr;                         // [n] token fed to the next step; = bos on entry
  int* eos = nullptr;                         // [n] row has ended; = 0 on entry
  const bf16_t* fcw = nullptr;                // [64, H] vocabulary projection (rows >= V zero)
  const float* fcb = nullptr;                 // [64]
  int V = 0, multinomial = 1, eos_id = 0, max_len = 0;
  float inv_temp = 1.0f;
  const float* uniforms = nullptr;            // pre-drawn: step i, row b at uniforms[(i-1) * u_stride + b]; null: Philox(seed, row_offset + row, step)
  long u_stride = 0;
  uint64_t seed = 0;
  long row_offset = 0;
  uint8_t* tokens = nullptr;                  // [n, max_len], initialised (bos, pad...) on entry
  int* end_pads = nullptr;                    // [n], = max_len on entry
  // parity hooks (pg_sample_forced): the token FED to step i+1 of row b is forced[b * max_len + i] instead of the drawn
  // one (the drawn one is still recorded); probs_out[((i-1) * p_stride + b) * 64 + v] receives softmax(y/temp) of step i
  const uint8_t* forced = nullptr;
  float* probs_out = nullptr;
  long p_stride = 0;
  int compact = 1;                            // ended rows leave the decode (tok = -1); 0: every row runs all T steps
  unsigned* sync = nullptr;                   // PG_SAMPLE_SYNC_WORDS words, zeroed by the launcher
  unsigned* fail = nullptr;                   // set by the launcher
  unsigned spin_limit = 0;                    // set by the launcher
};
bool dec_sample_persist_supported(int Hd, int L, int V, long n);
int launch_dec_sample_persist(DecSamplePersist& a, hipStream_t stream);
// PG_NO_PERSIST / pg_persist_disable mask (pg_vae.hip): bit 0 encoder, 1 decoder forward, 2 decoder BPTT, 3 sampler
int persist_off_mask();

bool dec_persist_supported(int H, int L, int B);
// the two-waves-per-SIMD forward kernel of the training configuration (pg_persist_dec8.hip), launched by launch_dec_fwd_persist
bool dec_fwd8_enabled();
int launch_dec_fwd8(DecFwdPersist& a, const Sched& sd, int grid, hipStream_t stream);
// host_flags: what the caller has taken over from the launcher, to keep it off the stream's critical path
constexpr int PG_PERSIST_SYNC_ZEROED = 1;   // a.sync was zeroed by the caller, ordered before this launch
constexpr int PG_PERSIST_NO_MIRROR = 2;     // the caller mirrors the give-up word to the host itself, later in the step
int launch_dec_fwd_persist(DecFwdPersist& a, const Sched& sd, int H, hipStream_t stream, int host_flags = 0);
// runs the whole decoder BPTT recurrence, then dh0_32 / dh0_op [B, H] = sum over layers of the h0 gradients
int launch_dec_bwd_persist(DecBwdPersist& a, const Sched& sd, int H, float* dh0_32, bf16_t* dh0_op, hipStream_t stream,
                           int host_flags = 0);
// process-wide sticky give-up word of the group-persistent kernels: device address (allocated on first use) and the
// stream-ordered copy to its pinned host mirror that persist_poll_failure() reads
int persist_fail_words(unsigned** dev);
int persist_fail_mirror(hipStream_t stream);
int persist_fail_refresh();   // blocking read of the device word into the host mirror (no-op before the first persistent launch)
bool persist_dma_plain();
unsigned long long* persist_stamps_for(int which);   // development aid: stamp buffer if tracing kernel `which`, else null
// nonzero once a persistent kernel of this process gave up waiting on another workgroup (checked by every entry point);
// reads the pinned mirror, which is current once the launching stream has been synchronised
int persist_poll_failure();
// device address of the sticky word, or null while no persistent kernel has been launched (pg_clip_adam hands it to the
// Adam kernel, which skips the update when it is set)
unsigned* persist_fail_dev_or_null();
// after a detected give-up: waits for the device, clears both words and switches every recurrence to the per-step kernels
int persist_recover();
// polls a waiting workgroup makes before it gives up (default 2^21, ~2-4 s); pg_persist_spin_limit() overrides it
unsigned persist_spin_limit();

// true when the persistent kernels cover this hidden size (others take the per-step path)
bool enc_persist_supported(int H);
int launch_enc_fwd_persist(const EncFwdPersist& a, const Sched& sd, int H, hipStream_t stream);
int launch_enc_bwd_persist(const EncBwdPersist& a, const Sched& sd, int H, hipStream_t stream);
