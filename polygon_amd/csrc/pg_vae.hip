// Entry points of the training path: shadow weights, encoder / decoder forward, BPTT, clip + Adam.
#include <algorithm>
#include <math.h>
#include "pg_gemm.h"
#include "pg_gru.h"
#include "pg_host.h"
#include "pg_layout.h"
#include "pg_misc.h"
#include "pg_mma.h"
#include "pg_persist.h"
#include <stdlib.h>
#include <type_traits>

thread_local char g_pg_err[512] = {0};

extern "C" const char* pg_last_error(void) { return g_pg_err; }
extern "C" int pg_version(void) { return PG_VERSION; }
extern "C" long pg_param_count(const pg_dims* d) { return d ? param_offsets(d).total : -1; }

// ===================================================================================================
// shadow weights
// ===================================================================================================
// ===================================================================================================
// measurement hook: event pairs around the two recurrences
// ===================================================================================================
namespace {
constexpr int PROF_MAX = 512;
struct ProfSide {
  hipEvent_t e0[PROF_MAX], e1[PROF_MAX];
  int created = 0, n = 0;
  long launches = 0;
  double flop = 0.0;
};
int g_prof_on = 0;
ProfSide g_prof[2];

// returns the slot to close with prof_end, or -1 when profiling is off / full
int prof_begin(int which, hipStream_t st) {
  if (!g_prof_on) return -1;
  ProfSide& p = g_prof[which];
  if (p.n >= PROF_MAX) return -1;
  if (p.n >= p.created) {
    if (hipEventCreate(&p.e0[p.created]) != hipSuccess || hipEventCreate(&p.e1[p.created]) != hipSuccess) return -1;
    ++p.created;
  }
  if (hipEventRecord(p.e0[p.n], st) != hipSuccess) return -1;
  return p.n;
}
void prof_end(int which, int slot, hipStream_t st, long launches, double flop) {
  if (slot < 0) return;
  ProfSide& p = g_prof[which];
  if (hipEventRecord(p.e1[slot], st) != hipSuccess) return;
  p.n = slot + 1; p.launches += launches; p.flop += flop;
}
}  // namespace

extern "C" int pg_profile_enable(int on) {
  g_prof_on = on ? 1 : 0;
  return PG_OK;
}

extern "C" int pg_profile_read(int which, double* total_ms, long* launches, double* flop, int* segments) {
  PG_REQUIRE(which == 0 || which == 1, "pg_profile_read: which must be 0 (forward) or 1 (backward)");
  ProfSide& p = g_prof[which];
  double ms = 0.0;
  for (int i = 0; i < p.n; ++i) {
    PG_CHECK_HIP(hipEventSynchronize(p.e1[i]));
    float m = 0.0f;
    PG_CHECK_HIP(hipEventElapsedTime(&m, p.e0[i], p.e1[i]));
    ms += m;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = p.launches;
  if (flop) *flop = p.flop;
  if (segments) *segments = p.n;
  p.n = 0; p.launches = 0; p.flop = 0.0;
  return PG_OK;
}

extern "C" size_t pg_shadow_bytes(const pg_dims* d, int dtype) {
  if (check_dims(d, dtype) != PG_OK) return 0;
  return shadow_offsets(d, dtype).total;
}

// (deferred-work plumbing, defined further down)
static int defer_fork(hipStream_t st, hipStream_t* side);
static int shadow_fork_done(hipStream_t side, hipStream_t st);

template <typename T>
static int prepare_weights_t(const pg_dims* d, int dtype, const float* params, void* shadow, hipStream_t st_main) {
  const ParamOff po = param_offsets(d);
  const ShadowOff so = shadow_offsets(d, dtype);
  const int Hq = d->Hq, Hd = d->Hd, Dz = d->Dz, E = d->E, V = d->V, Vp = vpad(d);
  // While weight-gradient deferral is on (the fused trainer), the DECODER's shadows are rebuilt on the library's side
  // stream, beside the encoder forward that follows (a 64-CU kernel that needs only the encoder's): pg_decoder_fwd /
  // pg_decoder_bwd / pg_sample join it on entry.  Otherwise everything runs on the caller's stream.
  hipStream_t sd = st_main;
  PG_TRY(defer_fork(st_main, &sd));
  PG_CHECK_HIP(hipMemsetAsync(at<char>(shadow, so.fc_w), 0, so.tab_enc - so.fc_w, sd));  // zero the padded fc rows
  PG_CHECK_HIP(hipMemsetAsync(at<char>(shadow, so.fc_b), 0, (size_t)Vp * 4, sd));
  CopyTable tabs[2];   // [0] encoder (caller's stream), [1] decoder
  long maxel[2] = {0, 0};
  tabs[0].n = tabs[1].n = 0;
  auto add = [&](int which, long src, size_t dst, int rows, int cols, int src_ld, int dst_ld, int tr, int f32) {
    CopyDesc c = {src, dst, rows, cols, src_ld, dst_ld, tr, f32};
    tabs[which].d[tabs[which].n++] = c;
    if ((long)rows * cols > maxel[which]) maxel[which] = (long)rows * cols;
  };
  const bool gen = enc_generic(d);   // other encoder shapes keep their own operand copies (pg_encgen.hip): nothing of theirs here
  if (!gen) {
  add(0, po.enc_whh, so.enc_whh, 3 * Hq, Hq, Hq, Hq, 0, 0);
  add(0, po.enc_whh, so.enc_whhT, 3 * Hq, Hq, Hq, 3 * Hq, 1, 0);
  add(0, po.qmu_w, so.wq, Dz, Hq, Hq, Hq, 0, 0);
  add(0, po.qlv_w, so.wq + (size_t)Dz * Hq * sizeof(T), Dz, Hq, Hq, Hq, 0, 0);
  add(0, po.qmu_w, so.wqT, Dz, Hq, Hq, 2 * Dz, 1, 0);
  add(0, po.qlv_w, so.wqT + (size_t)Dz * sizeof(T), Dz, Hq, Hq, 2 * Dz, 1, 0);
  add(0, po.qmu_b, so.bq, 1, Dz, Dz, Dz, 0, 1);
  add(0, po.qlv_b, so.bq + (size_t)Dz * 4, 1, Dz, Dz, Dz, 0, 1);
  }
  add(1, po.lat_w, so.lat_w, Hd, Dz, Dz, Dz, 0, 0);
  add(1, po.lat_w, so.lat_wT, Hd, Dz, Dz, Hd, 1, 0);
  add(1, po.dec_wih[0] + E, so.wih0z, 3 * Hd, Dz, E + Dz, Dz, 0, 0);
  add(1, po.dec_wih[0] + E, so.wih0zT, 3 * Hd, Dz, E + Dz, 3 * Hd, 1, 0);
  for (int l = 0; l < d->L; ++l) {
    add(1, po.dec_whh[l], so.whh[l], 3 * Hd, Hd, Hd, Hd, 0, 0);
    add(1, po.dec_whh[l], so.whhT[l], 3 * Hd, Hd, Hd, 3 * Hd, 1, 0);
    if (l > 0) {
      add(1, po.dec_wih[l], so.wih[l], 3 * Hd, Hd, Hd, Hd, 0, 0);
      add(1, po.dec_wih[l], so.wihT[l], 3 * Hd, Hd, Hd, 3 * Hd, 1, 0);
    }
  }
  add(1, po.fc_w, so.fc_w, V, Hd, Hd, Hd, 0, 0);
  add(1, po.fc_w, so.fc_wT, V, Hd, Hd, Vp, 1, 0);
  add(1, po.fc_b, so.fc_b, 1, V, V, V, 0, 1);
  PG_REQUIRE(tabs[0].n <= PG_MAX_COPY && tabs[1].n <= PG_MAX_COPY, "copy table overflow");
  // encoder part first, on the caller's stream: the encoder forward waits for nothing else
  if (!gen) {
    hipLaunchKernelGGL(copy_table_kernel<T>, dim3(std::min<int>(cdiv(maxel[0], 256), 256), tabs[0].n), dim3(256), 0, st_main, params,
                       static_cast<char*>(shadow), tabs[0]);
    PG_CHECK_LAUNCH();
    PG_TRY(launch_gi_table(params + po.x_emb, E, params + po.enc_wih, E, params + po.enc_bih, params + po.enc_bhh, Hq, V,
                           at<float>(shadow, so.tab_enc), st_main));
  }
  hipLaunchKernelGGL(copy_table_kernel<T>, dim3(std::min<int>(cdiv(maxel[1], 256), 256), tabs[1].n), dim3(256), 0, sd, params,
                     static_cast<char*>(shadow), tabs[1]);
  PG_CHECK_LAUNCH();
  PG_TRY(launch_gi_table(params + po.x_emb, E, params + po.dec_wih[0], E + Dz, params + po.dec_bih[0], params + po.dec_bhh[0], Hd, V,
                         at<float>(shadow, so.tab_dec), sd));
  for (int l = 1; l < d->L; ++l) {
    hipLaunchKernelGGL(bias_fold_kernel, dim3(cdiv(3 * Hd, 256)), dim3(256), 0, sd, params + po.dec_bih[l], params + po.dec_bhh[l],
                       Hd, at<float>(shadow, so.bias_gi[l]));
    PG_CHECK_LAUNCH();
  }
  PG_TRY(shadow_fork_done(sd, st_main));
  return PG_OK;
}

extern "C" int pg_prepare_weights(const pg_dims* d, int dtype, const float* params, void* shadow, void* stream) {
  PG_TRY(check_dims(d, dtype));
  PG_REQUIRE(params && shadow, "pg_prepare_weights: NULL buffer");
  return dtype == PG_BF16 ? prepare_weights_t<bf16_t>(d, dtype, params, shadow, (hipStream_t)stream)
                          : prepare_weights_t<float>(d, dtype, params, shadow, (hipStream_t)stream);
}

// ===================================================================================================
// workspace
// ===================================================================================================
extern "C" size_t pg_train_workspace_bytes(const pg_dims* d, int dtype, int B, long n_tokens) {
  if (check_dims(d, dtype) != PG_OK || B < 1 || n_tokens < B) return 0;
  return train_ws_layout(d, dtype, B, n_tokens, n_tokens).total;
}

static inline long sum_len(const int32_t* l, int B) {
  long s = 0;
  for (int i = 0; i < B; ++i) s += l[i];
  return s;
}

// The persistent recurrence kernels (pg_persist.h) run unless the size is not covered.  PG_NO_PERSIST=<mask> (environment,
// read once) or pg_persist_disable(mask) force the per-step kernels: bit 0 encoder, bit 1 decoder forward, bit 2 decoder
// BPTT, bit 3 the sampler (A/B timing, and a second implementation for the parity tests).
static int g_persist_off = -1;
int persist_off_mask() {
  if (g_persist_off < 0) g_persist_off = getenv("PG_NO_PERSIST") ? atoi(getenv("PG_NO_PERSIST")) & 15 : 0;
  return g_persist_off;
}
extern "C" int pg_persist_disable(int mask) {
  const int old = persist_off_mask();
  if (mask >= 0) g_persist_off = mask & 15;
  return old;
}
static bool use_persistent(int H) { return !(persist_off_mask() & 1) && enc_persist_supported(H); }

// A persistent kernel of an earlier call gave up waiting for a peer workgroup (its grid was not co-resident: GPU shared or
// partially available).  Everything computed since that launch is suspect - the Adam kernel has skipped its updates on
// seeing the device word - so THIS call fails, loudly, once; the library then clears the word and switches every
// recurrence to the per-step kernels, which need no co-residency, so later calls work (pg_persist_disable(0) re-enables).
static int g_entry_checks = 1;
static int check_persist_now();
int check_persist() { return g_entry_checks ? check_persist_now() : PG_OK; }
// Entry points poll the give-up word's mirror when they are called, i.e. whenever the device happens to have finished
// the failing kernel: under data parallelism a rank that raised in the MIDDLE of a step would leave its peers waiting in
// the step's all-reduce.  The data-parallel trainer therefore switches the entry checks off for the epoch and relies on
// the shared flag (pg_fail_flag_export / _import: every rank skips the update) and on pg_check_failure at the epoch's end,
// which every rank reaches.  Returns the previous setting (on < 0: query).
extern "C" int pg_entry_checks(int on) {
  const int old = g_entry_checks;
  if (on >= 0) g_entry_checks = on ? 1 : 0;
  return old;
}
static int check_persist_now() {
  if (persist_poll_failure() == 0) return PG_OK;
  PG_TRY(persist_recover());
  g_persist_off = 15;
  PG_FAIL(PG_ERR_PERSIST,
          "a persistent recurrence kernel timed out waiting for its peer workgroups (GPU shared or partially available?): "
          "the results since that launch are invalid and optimiser updates were skipped; later calls use the per-step kernels");
}

extern "C" int pg_check_failure(void* stream) {
  PG_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  PG_TRY(persist_fail_refresh());   // the device word itself: the asynchronous mirrors are an early warning only
  return check_persist_now();
}

// Data parallelism: the give-up word is per process, but a rank whose recurrence gave up has already SUM-all-reduced its
// garbage gradients into every other rank.  The trainer therefore exports the word behind the step's kernels, MAX-all-
// reduces it with the gradients and imports the result BEFORE pg_clip_adam: every rank's Adam kernel then skips the update
// (weights stay identical across ranks) and every rank's next pg_check_failure raises (nobody saves, nobody is left
// waiting in a collective).
extern "C" int pg_fail_flag_export(int32_t* dst, void* stream) {
  PG_REQUIRE(dst, "pg_fail_flag_export: NULL buffer");
  unsigned* dev = nullptr;
  PG_TRY(persist_fail_words(&dev));
  PG_CHECK_HIP(hipMemcpyAsync(dst, dev, sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return PG_OK;
}
extern "C" int pg_fail_flag_import(const int32_t* src, void* stream) {
  PG_REQUIRE(src, "pg_fail_flag_import: NULL buffer");
  unsigned* dev = nullptr;
  PG_TRY(persist_fail_words(&dev));
  PG_CHECK_HIP(hipMemcpyAsync(dev, src, sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return persist_fail_mirror((hipStream_t)stream);
}

// ---- deferred weight gradients: the decoder's large weight-gradient GEMMs run on a library-owned side stream, beside the
// encoder BPTT (a row-resident kernel on 64 of the 256 CUs, latency-bound).  Never beside the group-persistent decoder
// kernels, whose workgroups must all be co-resident.  Joined at the end of pg_encoder_bwd, in pg_clip_adam and by
// pg_deferred_join.
static int g_defer = 0;
static int g_defer_dev = -1;
static hipStream_t g_side = nullptr, g_side2 = nullptr;   // large GEMMs / small latency-bound tail
static hipEvent_t g_ev_fork = nullptr, g_ev_join = nullptr, g_ev_join2 = nullptr;
static bool g_defer_pending = false;
static hipEvent_t g_ev_shadow = nullptr;     // decoder shadows rebuilt on the side stream (pg_prepare_weights)
static bool g_shadow_pending = false;
// The batch the encoder forward was given while that rebuild was pending: an event on the caller's stream behind the
// encoder's packing kernel (the tokens are complete there, and nothing earlier on the stream still uses the workspace),
// which the side stream waits for.  The encoder's token sort, and the prologue of the decoder forward of the SAME tokens and
// workspace, may then run on the side stream.
static hipEvent_t g_ev_tok = nullptr;
static const void* g_tok_ptr = nullptr;
static const void* g_tok_ws = nullptr;
static int note_tokens(const void* d_tokens, const void* ws, hipStream_t st) {
  g_tok_ptr = nullptr;
  if (!(g_shadow_pending && g_side && !g_defer_pending)) return PG_OK;
  if (!g_ev_tok) PG_CHECK_HIP(hipEventCreateWithFlags(&g_ev_tok, hipEventDisableTiming));
  PG_CHECK_HIP(hipEventRecord(g_ev_tok, st));
  PG_CHECK_HIP(hipStreamWaitEvent(g_side, g_ev_tok, 0));
  g_tok_ptr = d_tokens;
  g_tok_ws = ws;
  return PG_OK;
}
extern "C" int pg_defer_weight_grads(int on) {
  const int old = g_defer;
  if (on >= 0) g_defer = on ? 1 : 0;
  return old;
}
static int defer_fork(hipStream_t st, hipStream_t* side) {   // *side = stream for the deferred GEMMs (st itself when off)
  *side = st;
  if (!g_defer) return PG_OK;
  int dev = 0;
  PG_CHECK_HIP(hipGetDevice(&dev));
  if (!g_side) {
    // default priority: a lowest-priority side stream was worth 15 us per step, but with two processes on one GPU (the
    // two-rank test) the low-priority queue was intermittently never scheduled and the step hung
    PG_CHECK_HIP(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
    PG_CHECK_HIP(hipStreamCreateWithFlags(&g_side2, hipStreamNonBlocking));
    PG_CHECK_HIP(hipEventCreateWithFlags(&g_ev_join2, hipEventDisableTiming));
    PG_CHECK_HIP(hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming));
    PG_CHECK_HIP(hipEventCreateWithFlags(&g_ev_join, hipEventDisableTiming));
    PG_CHECK_HIP(hipEventCreateWithFlags(&g_ev_shadow, hipEventDisableTiming));
    g_defer_dev = dev;
  }
  if (dev != g_defer_dev) return PG_OK;   // one process per GPU is the model; another device simply does not defer
  PG_CHECK_HIP(hipEventRecord(g_ev_fork, st));
  PG_CHECK_HIP(hipStreamWaitEvent(g_side, g_ev_fork, 0));
  PG_CHECK_HIP(hipStreamWaitEvent(g_side2, g_ev_fork, 0));
  *side = g_side;
  return PG_OK;
}
// the second side stream of the fork just made (the caller's stream itself when not deferring)
static hipStream_t defer_second(hipStream_t side, hipStream_t st) { return side == st ? st : g_side2; }
// (A join is not repeated on a stream that has already made it: every wait is a barrier packet on the caller's stream, ~6 us
// each even when the event has long been signalled, and the step used to end on four of them.  Chaining the two side
// streams into one event was tried: the encoder's embedding tail on the second then queues behind the first.)
static hipStream_t g_joined_on = nullptr;
static bool g_joined = false;   // (the legacy default stream is a null handle: a flag, not a sentinel)
static int defer_done(hipStream_t side, hipStream_t st) {
  if (side == st) return PG_OK;
  PG_CHECK_HIP(hipEventRecord(g_ev_join, side));
  PG_CHECK_HIP(hipEventRecord(g_ev_join2, g_side2));
  g_defer_pending = true;
  g_joined = false;
  return PG_OK;
}
// more work was queued on the second side stream since defer_done (the encoder's embedding tail): joins wait for this record
static int defer_rerecord() {
  PG_CHECK_HIP(hipEventRecord(g_ev_join2, g_side2));
  g_joined = false;
  return PG_OK;
}
static int defer_join(hipStream_t st, bool clear) {
  if (g_defer_pending) {
    if (!(g_joined && g_joined_on == st)) {
      PG_CHECK_HIP(hipStreamWaitEvent(st, g_ev_join, 0));
      PG_CHECK_HIP(hipStreamWaitEvent(st, g_ev_join2, 0));
    }
    g_joined = true;
    g_joined_on = st;
    if (clear) g_defer_pending = false;
  }
  return PG_OK;
}
extern "C" int pg_deferred_join(void* stream) { return defer_join((hipStream_t)stream, false); }
static int shadow_fork_done(hipStream_t side, hipStream_t st) {
  if (side == st) return PG_OK;
  PG_CHECK_HIP(hipEventRecord(g_ev_shadow, side));
  g_shadow_pending = true;
  return PG_OK;
}
// every consumer of the decoder's shadow weights calls this first
int shadow_join(hipStream_t st) {
  if (g_shadow_pending) {
    PG_CHECK_HIP(hipStreamWaitEvent(st, g_ev_shadow, 0));
    g_shadow_pending = false;
  }
  return PG_OK;
}

// the lane-order form of the saved gate activations packs (r, z) into d_sv[l][0..1] and (n, hn) into d_sv[l][2..3]: the
// two arrays of a pair must be one contiguous region (they are, by construction of train_ws_layout; checked, not assumed)
static bool sv_pairs_adjacent(const TrainWs& w, int L) {
  for (int l = 0; l < L; ++l) {
    const size_t sz = w.d_sv[l][1] - w.d_sv[l][0];
    if (w.d_sv[l][3] - w.d_sv[l][2] != sz || w.d_sv[l][2] - w.d_sv[l][1] != sz) return false;
  }
  return true;
}
static bool use_persistent_dec(int H, int L, int B) { return !(persist_off_mask() & 2) && dec_persist_supported(H, L, B); }
static bool use_persistent_dec_bwd(int H, int L, int B) { return !(persist_off_mask() & 4) && dec_persist_supported(H, L, B); }

// Layout of the saved gate activations r, z, n, hn in a training workspace: packed rows (0) or the persistent kernels'
// lane order (1).  pg_decoder_fwd RECORDS what it wrote, per workspace, and pg_decoder_bwd reads the record instead of
// deriving the layout again from the persist mask current at ITS call: the mask can change in between (a detected
// give-up switches every recurrence to the per-step kernels; pg_persist_disable is an A/B hook), and a BPTT that reads
// lane-ordered activations as packed rows produces silently wrong gradients.
struct SvLayoutRec { const void* ws; int lane; };
static SvLayoutRec g_sv_layout[64] = {};
static int g_sv_layout_next = 0;
// Whether the BPTT counters of a workspace are still as pg_decoder_fwd's prologue zeroed them (set = true), consumed by
// the first pg_decoder_bwd after it (set = false; returns what it found).  Same stream order as the saved activations.
static const void* g_bwd_sync_clean[8] = {};
static bool bwd_sync_record(const void* ws, bool set) {
  static int next = 0;
  for (auto& r : g_bwd_sync_clean)
    if (r == ws) { if (!set) r = nullptr; return true; }
  if (set) { g_bwd_sync_clean[next] = ws; next = (next + 1) % 8; }
  return false;
}
// Whether the workspace holds the rows grouped by token (pg_scatter.h) for the batch its last forward pass packed: bit 0
// encoder, bit 1 decoder.  Written by the forward passes (set / clear), read by the backward passes.
struct TokSortRec { const void* ws; int mask; };
static TokSortRec g_tok_sorted[8] = {};
static void tok_sorted_set(const void* ws, int bit, bool on) {
  static int next = 0;
  for (auto& r : g_tok_sorted)
    if (r.ws == ws) { r.mask = on ? (r.mask | bit) : (r.mask & ~bit); return; }
  if (!on) return;
  g_tok_sorted[next] = {ws, bit};
  next = (next + 1) % 8;
}
static bool tok_sorted(const void* ws, int bit) {
  for (auto& r : g_tok_sorted)
    if (r.ws == ws) return (r.mask & bit) != 0;
  return false;
}
static_assert(PG_TS_ROWS >= 128 && PG_TS_ROWS % 128 == 0, "TrainWs::d_ts / e_ts / e_slab are sized for chunks of at least 128 rows (pg_layout.h)");
static TokSort tok_sort_at(void* ws, const size_t (&o)[3]) {
  TokSort ts;
  ts.perm = at<int>(ws, o[0]); ts.chunks = at<int4>(ws, o[1]); ts.tokchunk = at<int>(ws, o[2]);
  return ts;
}
static void sv_layout_record(const void* ws, int lane) {
  for (auto& r : g_sv_layout)
    if (r.ws == ws) { r.lane = lane; return; }
  g_sv_layout[g_sv_layout_next] = SvLayoutRec{ws, lane};
  g_sv_layout_next = (g_sv_layout_next + 1) % 64;
}
static int sv_layout_of(const void* ws) {   // -1: no forward pass with save_for_backward ran on this workspace
  for (const auto& r : g_sv_layout)
    if (r.ws == ws) return r.lane;
  return -1;
}

// ===================================================================================================
// encoder forward
// ===================================================================================================
template <typename T>
static int encoder_fwd_t(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes, int B,
                         int Tmax, const int32_t* d_tokens, const int32_t* lens, const float* eps, const pg_dp* dp, int save,
                         float* z, float* mu, float* logvar, float* kl, hipStream_t st) {
  const ParamOff po = param_offsets(d);
  const ShadowOff so = shadow_offsets(d, dtype);
  Sched se;
  PG_TRY(make_sched(lens, B, Tmax, 0, &se));
  const long Ne = se.off[se.T];
  const TrainWs w = train_ws_layout(d, dtype, B, Ne, Ne);
  PG_REQUIRE(ws_bytes >= w.total, "encoder_fwd: workspace too small (%zu < %zu)", ws_bytes, w.total);
  const int Hq = d->Hq, Dz = d->Dz;
  float* scal = at<float>(ws, w.scalars);
  hipLaunchKernelGGL(build_packed_kernel<T>, dim3(se.T, cdiv(B, 256)), dim3(256), 0, st, d_tokens, B, Tmax, d->pad, se, 0,
                     at<int>(ws, w.enc_tok), (int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)nullptr,
                     reinterpret_cast<unsigned*>(scal + SC_KL_SUM), 2);   // (zeroes the KL sum and its ticket)
  PG_CHECK_LAUNCH();
  // The rows grouped by token, for the embedding-table gradient of the backward pass (pg_scatter.h): one block, needs the
  // packed tokens only.  Beside the recurrence on the library's side stream while the decoder's shadow rebuild is pending
  // there (an event behind the packing kernel, which the side stream waits for; the join is the one the decoder forward
  // owes anyway, re-recorded behind the sort), else in order.
  PG_TRY(note_tokens(d_tokens, ws, st));
  static const bool sort_in_bwd = getenv("PG_SORT_IN_BWD") != nullptr;   // A/B aid: leave the sort to the backward pass
  const bool sort_now = save && !sort_in_bwd && Ne <= tok_sort_max_rows(vpad(d));
  if (sort_now) {
    hipStream_t ss = g_tok_ptr ? g_side : st;
    PG_TRY(launch_tok_sort(at<int>(ws, w.enc_tok), Ne, vpad(d), tok_sort_at(ws, w.e_ts), ss));
    if (ss != st) PG_CHECK_HIP(hipEventRecord(g_ev_shadow, g_side));
  }
  tok_sorted_set(ws, 1, sort_now);
  float* carry = at<float>(ws, w.e_carry);
  bool persist = false;
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (use_persistent(Hq)) {   // one launch walks every step with the recurrent weights resident on the CU
      EncFwdPersist pa;
      pa.Whh = at<T>(shadow, so.enc_whh); pa.table = at<float>(shadow, so.tab_enc); pa.bhn = params + po.enc_bhh + 2 * Hq;
      pa.tok = at<int>(ws, w.enc_tok);
      if (save) {
        pa.sv_r = at<T>(ws, w.e_sv[0]); pa.sv_z = at<T>(ws, w.e_sv[1]); pa.sv_n = at<T>(ws, w.e_sv[2]);
        pa.sv_hn = at<T>(ws, w.e_sv[3]); pa.sv_hprev = at<T>(ws, w.e_sv[4]);
      }
      pa.h_fin32 = carry; pa.h_fin_op = at<T>(ws, w.e_hfin_op); pa.B = B;
      PG_TRY(launch_enc_fwd_persist(pa, se, Hq, st));
      persist = true;
    }
  }
  for (int t = 0; t < se.T && !persist; ++t) {
    GruFwdArgs<T> a;
    a.rows = sched_B(se, t); a.H = Hq;
    if (t > 0) { a.hprev_op = at<T>(ws, w.e_hop[(t - 1) & 1]); a.hprev32 = carry; }
    a.Whh = at<T>(shadow, so.enc_whh);
    a.bhn = params + po.enc_bhh + 2 * Hq;
    a.gi_mode = PG_GI_TABLE;
    a.table = at<float>(shadow, so.tab_enc);
    a.tok = at<int>(ws, w.enc_tok) + se.off[t];
    a.h_op = at<T>(ws, w.e_hop[t & 1]);
    a.h32 = carry;
    if (save) {
      const long o = (long)se.off[t] * Hq;
      a.sv_r = at<T>(ws, w.e_sv[0]) + o; a.sv_z = at<T>(ws, w.e_sv[1]) + o; a.sv_n = at<T>(ws, w.e_sv[2]) + o;
      a.sv_hn = at<T>(ws, w.e_sv[3]) + o; a.sv_hprev = at<T>(ws, w.e_sv[4]) + o;
    }
    PG_TRY(launch_gru_fwd<T>(a, st));
  }
  // q_mu | q_logvar on the final states (each row's own last step: the carry is updated in place)
  if (!persist) PG_TRY(convert_to<T>(carry, at<T>(ws, w.e_hfin_op), (long)B * Hq, st));
  const float inv_B = (dp && dp->kl_inv_batch > 0.0f) ? dp->kl_inv_batch : 1.0f / (float)B;
  if constexpr (std::is_same<T, bf16_t>::value) {   // heads + reparameterisation + KL as one kernel (pg_heads.hip)
    int status = PG_OK;
    if (launch_enc_heads(at<T>(ws, w.e_hfin_op), B, Hq, at<T>(shadow, so.wq), at<float>(shadow, so.bq), Dz, eps, at<float>(ws, w.e_mulv), z,
                         at<T>(ws, w.e_zop), mu, logvar, save ? at<float>(ws, w.e_eps) : (float*)nullptr, scal + SC_KL_SUM, inv_B, kl, st,
                         &status))
      return status;
  }
  GemmEpi<T> e;
  e.bias = at<float>(shadow, so.bq); e.C32 = at<float>(ws, w.e_mulv); e.ldc = 2 * Dz;
  PG_TRY(launch_gemm_nt<T>(at<T>(ws, w.e_hfin_op), Hq, at<T>(shadow, so.wq), Hq, B, 2 * Dz, Hq, e, st));
  hipLaunchKernelGGL(reparam_kl_kernel<T>, dim3(std::min(cdiv((long)B * Dz, 256), 64)), dim3(256), 0, st, at<float>(ws, w.e_mulv), eps, B, Dz, z,
                     at<T>(ws, w.e_zop), mu, logvar, save ? at<float>(ws, w.e_eps) : (float*)nullptr, scal + SC_KL_SUM, inv_B, kl);
  PG_CHECK_LAUNCH();
  return PG_OK;
}

extern "C" int pg_encoder_fwd(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes,
                              int B, int Tmax, const int32_t* d_tokens, const int32_t* h_lengths, const float* eps,
                              const pg_dp* dp, int save_for_backward, float* z, float* mu, float* logvar, float* kl,
                              void* stream) {
  PG_TRY(check_dims(d, dtype));
  PG_TRY(check_persist());
  PG_REQUIRE(params && shadow && ws && d_tokens && z, "pg_encoder_fwd: NULL buffer");
  PG_REQUIRE(!enc_generic(d), "pg_encoder_fwd serves the one-layer one-direction encoder; Lq=%d qbidir=%d goes to pg_encoder_gen_fwd",
             d->Lq, d->qbidir);
  return dtype == PG_BF16
             ? encoder_fwd_t<bf16_t>(d, dtype, params, shadow, ws, ws_bytes, B, Tmax, d_tokens, h_lengths, eps, dp,
                                     save_for_backward, z, mu, logvar, kl, (hipStream_t)stream)
             : encoder_fwd_t<float>(d, dtype, params, shadow, ws, ws_bytes, B, Tmax, d_tokens, h_lengths, eps, dp,
                                    save_for_backward, z, mu, logvar, kl, (hipStream_t)stream);
}

// ===================================================================================================
// decoder forward
// ===================================================================================================
// Fused vocabulary projection + log-softmax + cross entropy + dlogits (vae_model.py:287-293: decoder_fc, then
// F.cross_entropy over the packed rows with ignore_index = pad).  One workgroup = 128 packed rows: logits tile on MFMA
// (W_fc is 64 padded rows: the whole vocabulary is one N tile), parked in LDS, then one wave per row does what ce_kernel
// does - max / sum-exp by wave reductions, loss and count accumulated per block (two atomics per block), softmax - onehot
// written as the operand of the backward GEMMs.  The fp32 logits are still written (the API hands them out); nothing
// reads them back on the training path.
template <class Cfg>
__global__ __launch_bounds__(Cfg::NTHREADS) void fc_ce_kernel(const typename Cfg::T* __restrict__ htop, long N, int H,
                                                               const typename Cfg::T* __restrict__ fcw,
                                                               const float* __restrict__ fcb, int V, int Vp,
                                                               const int* __restrict__ tgt, int pad,
                                                               float* __restrict__ logits, typename Cfg::T* __restrict__ dlogits,
                                                               float* __restrict__ scal, float count_override,
                                                               float* __restrict__ recon_out) {
  using T = typename Cfg::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int m0 = blockIdx.x * Cfg::BM;
  f32x16 acc[Cfg::TM][Cfg::TN];
  {
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = zacc;
  }
  RowSrc<T> as{htop, H, m0, (int)N};
  RowSrc<T> bs{fcw, H, 0, Cfg::BN};
  mma_mainloop_nt<Cfg, Cfg::TN, 0>(acc, as, bs, H, smem);
  float* sl = reinterpret_cast<float*>(smem);  // [BM][65]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
#pragma unroll
  for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < Cfg::TN; ++tn)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        sl[Cfg::row_of(wm, tm, reg, lane) * 65 + Cfg::col_in_group(wn, tn, lane & 31)] = acc[tm][tn][reg];
  __syncthreads();
  constexpr int NW = Cfg::NTHREADS / 64;
  const float bias = lane < Vp ? fcb[lane] : 0.0f;
  float loss = 0.0f, cnt = 0.0f;
  // the targets of this wave's rows in ONE load (lane k: the row of iteration k): a load per iteration put a memory round
  // trip on every row
  static_assert(Cfg::BM / NW <= 64, "one lane per iteration");
  const long my_row = (long)m0 + wave + lane * NW;
  const int tg_all = (lane < Cfg::BM / NW && my_row < N) ? tgt[my_row] : pad;
  for (int r = wave, it = 0; r < Cfg::BM; r += NW, ++it) {
    const long row = (long)m0 + r;
    if (row >= N) break;
    const float y = sl[r * 65 + lane] + bias;
    if (logits && lane < Vp) logits[row * Vp + lane] = y;
    const float x = lane < V ? y : -INFINITY;
    const float mx = wave_max(x);
    const float e = lane < V ? __expf(x - mx) : 0.0f;
    const float s = wave_sum(e);
    const int tg = __builtin_amdgcn_readlane(tg_all, it);
    const bool w = tg != pad;
    const float xt = __shfl(x, tg < 64 ? tg : 0, 64);
    if (w) { loss += __logf(s) + mx - xt; cnt += 1.0f; }
    if (dlogits && lane < Vp) {
      const float p = e / s;
      dlogits[row * Vp + lane] = from_f32<T>(w && lane < V ? p - (lane == tg ? 1.0f : 0.0f) : 0.0f);
    }
  }
  __syncthreads();   // the logits tile is dead: its first words carry the per-wave partial sums
  if (lane == 0) { sl[wave] = loss; sl[16 + wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, c = 0.0f;
    for (int w = 0; w < NW; ++w) { a += sl[w]; c += sl[16 + w]; }
    ce_add_and_finalize(scal, a, c, gridDim.x, count_override, recon_out);
  }
}

template <typename T>
static int decoder_fwd_t(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes, int B,
                         int Tmax, const int32_t* d_tokens, const int32_t* lens, const float* z, const pg_dp* dp, float p_drop,
                         uint64_t seed, int all_steps, int save, float* recon, float** logits_packed, hipStream_t st) {
  const ParamOff po = param_offsets(d);
  const ShadowOff so = shadow_offsets(d, dtype);
  Sched sd;
  // (on an argument error the shadow join is still owed: the early returns below go through it)
  struct JoinOnExit { hipStream_t s; bool done; ~JoinOnExit() { if (!done) shadow_join(s); } } join_guard{st, false};
  PG_TRY(make_sched(lens, B, Tmax, all_steps ? 0 : 1, &sd));
  PG_REQUIRE(sd.T >= 1, "decoder_fwd: sequences of length 1 have no decoder step");
  const long Ne = sum_len(lens, B), Nd = sd.off[sd.T];
  const TrainWs w = train_ws_layout(d, dtype, B, Ne, Ne);
  PG_REQUIRE(ws_bytes >= w.total, "decoder_fwd: workspace too small (%zu < %zu)", ws_bytes, w.total);
  PG_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "decoder_fwd: dropout %f outside [0,1)", p_drop);
  const int Hd = d->Hd, Dz = d->Dz, L = d->L, Vp = vpad(d);
  const int b_mul = dp ? dp->b_mul : 1, b_add = dp ? dp->b_add : 0;
  // The packing of the batch and the zeroing of the loss words need nothing of the encoder: while the decoder's shadow
  // weights of this step are being rebuilt on the library's side stream (pg_prepare_weights under deferral, beside the
  // encoder forward), they go there too, behind the shadows, and the caller's stream picks both up with the ONE event wait
  // it owes anyway - 12 us of launches and a second wait off the stretch between the two forward recurrences.  (Not while
  // the deferred tail of an earlier micro-batch may still read this workspace's packed tokens and token tables on the second
  // side stream: g_defer_pending.)
  const bool side_pro = g_shadow_pending && g_side && !g_defer_pending && g_tok_ptr == d_tokens && g_tok_ws == ws;
  g_tok_ptr = nullptr;
  hipStream_t pro = side_pro ? g_side : st;
  // zeroed by the same kernel, as ONE span: the counters of the forward recurrence and of this step's BPTT, the two CE sums
  // and their ticket
  float* scal = at<float>(ws, w.scalars);
  static_assert(SC_CE_SUM == 0 && SC_CE_TICKET == 2, "the CE words lead the scalars");
  PG_REQUIRE(w.sync < w.sync_bwd && w.sync_bwd < w.scalars && sd.T >= 1, "decoder_fwd: layout of the zeroed span");
  hipLaunchKernelGGL(build_packed_kernel<T>, dim3(sd.T, cdiv(B, 256)), dim3(256), 0, pro, d_tokens, B, Tmax, d->pad, sd, 1,
                     at<int>(ws, w.dec_tok), at<int>(ws, w.dec_tgt), at<int>(ws, w.dec_t), at<int>(ws, w.dec_b),
                     at<int>(ws, w.dec_off), at<unsigned>(ws, w.sync),
                     (int)((w.scalars - w.sync) / 4 + 3));
  PG_CHECK_LAUNCH();
  bwd_sync_record(ws, true);
  // the rows grouped by token, for the embedding-table gradient of the backward pass: needs the tokens only
  static const bool sort_in_bwd = getenv("PG_SORT_IN_BWD") != nullptr;   // A/B aid: leave the sort to the backward pass
  const bool sort_now = save && !sort_in_bwd && Nd <= tok_sort_max_rows(Vp);
  if (sort_now) PG_TRY(launch_tok_sort(at<int>(ws, w.dec_tok), Nd, Vp, tok_sort_at(ws, w.d_ts), pro));
  tok_sorted_set(ws, 2, sort_now);
  if (pro != st) PG_CHECK_HIP(hipEventRecord(g_ev_shadow, pro));               // (re-recorded behind the prologue)
  join_guard.done = true;
  PG_TRY(shadow_join(st));
  T* z_op = at<T>(ws, w.d_zop);
  bool zin_done = false;
  if constexpr (std::is_same<T, bf16_t>::value) {   // z -> bf16, z projection and h0 as one kernel (pg_heads.hip)
    int status = PG_OK;
    zin_done = launch_dec_zin(z, B, Dz, Hd, at<T>(shadow, so.wih0z), at<T>(shadow, so.lat_w), params + po.lat_b, z_op,
                              at<float>(ws, w.d_zproj), at<float>(ws, w.d_h0_32), at<T>(ws, w.d_h0_op), st, &status);
    PG_TRY(status);
  }
  if (!zin_done) {
    PG_TRY(convert_to<T>(z, z_op, (long)B * Dz, st));
    {  // per-sequence half of the layer-0 input projection: W_ih0[:, E:] z
      GemmEpi<T> e;
      e.C32 = at<float>(ws, w.d_zproj); e.ldc = 3 * Hd;
      PG_TRY(launch_gemm_nt<T>(z_op, Dz, at<T>(shadow, so.wih0z), Dz, B, 3 * Hd, Dz, e, st));
    }
    {  // h0 = decoder_lat(z)
      GemmEpi<T> e;
      e.bias = params + po.lat_b; e.C32 = at<float>(ws, w.d_h0_32); e.Ct = at<T>(ws, w.d_h0_op); e.ldc = Hd;
      PG_TRY(launch_gemm_nt<T>(z_op, Dz, at<T>(shadow, so.lat_w), Dz, B, Hd, Dz, e, st));
    }
  }
  const bool drop = p_drop > 0.0f;
  const bool persist_dec = std::is_same<T, bf16_t>::value && use_persistent_dec(Hd, L, B);
  for (int l = 0; l < L && !persist_dec; ++l)   // (the persistent kernel reads h0 itself at step 0)
    PG_CHECK_HIP(hipMemcpyAsync(at<float>(ws, w.d_carry[l]), at<float>(ws, w.d_h0_32), (size_t)B * Hd * 4,
                                hipMemcpyDeviceToDevice, st));
  // Wavefront over the (layer, t) grid: cell (l, t) needs (l, t-1) and (l-1, t), so all cells of one anti-diagonal
  // are independent and run in ONE launch; layers >= 1 take their input projection fused (x = output of the layer
  // below at the same step).  T + L - 1 sequential launches instead of T * L.
  bool persist = false;
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (use_persistent_dec(Hd, L, B)) {   // ONE launch: weights stationary per workgroup, layers pipelined through counters
      static_assert(PG_PERSIST_SYNC_WORDS == 2064, "TrainWs::sync is sized for 2064 words");
      DecFwdPersist pa;
      pa.L = L; pa.B = B;
      for (int l = 0; l < L; ++l) {
        pa.Whh[l] = at<T>(shadow, so.whh[l]);
        pa.bhn[l] = params + po.dec_bhh[l] + 2 * Hd;
        if (l > 0) { pa.Wih[l] = at<T>(shadow, so.wih[l]); pa.bias_gi[l] = at<float>(shadow, so.bias_gi[l]); }
        pa.hseq[l] = at<T>(ws, w.d_hseq[l]);
        pa.hdrop[l] = (drop && l < L - 1) ? at<T>(ws, w.d_hdrop[l]) : nullptr;
        pa.carry[l] = at<float>(ws, w.d_carry[l]);
        if (save) for (int i = 0; i < 5; ++i) pa.sv[l][i] = at<T>(ws, w.d_sv[l][i]);
      }
      pa.sv_lane = (use_persistent_dec_bwd(Hd, L, B) && sv_pairs_adjacent(w, L)) ? 1 : 0;   // the BPTT of this step will be the persistent kernel too
      pa.table = at<float>(shadow, so.tab_dec); pa.tok = at<int>(ws, w.dec_tok); pa.zproj = at<float>(ws, w.d_zproj);
      pa.h0_op = at<T>(ws, w.d_h0_op); pa.h0_32 = at<float>(ws, w.d_h0_32);
      pa.p_drop = p_drop; pa.seed = seed; pa.b_mul = b_mul; pa.b_add = b_add;
      pa.sync = at<unsigned>(ws, w.sync);
      const int pslot = prof_begin(0, st);
      // (counters zeroed by the prologue; the training forward leaves the host mirror of the give-up word to the BPTT of
      // the same step - the word is sticky, the Adam kernel reads it on the device, pg_check_failure reads it itself)
      const bool bptt_next = save && pa.sv_lane != 0;
      PG_TRY(launch_dec_fwd_persist(pa, sd, Hd, st, PG_PERSIST_SYNC_ZEROED | (bptt_next ? PG_PERSIST_NO_MIRROR : 0)));
      if (save) sv_layout_record(ws, pa.sv_lane);   // what the launcher's kernel wrote: 1, or 2 (the two-waves-per-SIMD kernel's order)
      double pflop = 0.0;
      for (int t = 0; t < sd.T; ++t) pflop += 2.0 * sched_B(sd, t) * 3.0 * Hd * (L * Hd + (L - 1) * Hd);
      prof_end(0, pslot, st, 1, pflop);
      persist = true;
    }
  }
  if (!persist && save) sv_layout_record(ws, 0);
  const int pslot = persist ? -1 : prof_begin(0, st);
  double pflop = 0.0;
  for (int dgi = 0; dgi < sd.T + L - 1 && !persist; ++dgi) {
    GruFwdDiag<T> dg;
    for (int l = 0; l < L; ++l) {
      const int t = dgi - l;
      if (t < 0 || t >= sd.T) continue;
      GruFwdArgs<T>& a = dg.a[dg.n++];
      T* hseq = at<T>(ws, w.d_hseq[l]);
      const long o = (long)sd.off[t] * Hd;
      a.rows = sched_B(sd, t); a.H = Hd;
      a.hprev_op = t == 0 ? at<T>(ws, w.d_h0_op) : hseq + (long)sd.off[t - 1] * Hd;
      a.hprev32 = at<float>(ws, w.d_carry[l]);
      a.Whh = at<T>(shadow, so.whh[l]);
      a.bhn = params + po.dec_bhh[l] + 2 * Hd;
      if (l == 0) {
        a.gi_mode = PG_GI_TABLE;
        a.table = at<float>(shadow, so.tab_dec);
        a.tok = at<int>(ws, w.dec_tok) + sd.off[t];
        a.zproj = at<float>(ws, w.d_zproj);
      } else {
        a.gi_mode = PG_GI_FUSED;
        a.x = (drop ? at<T>(ws, w.d_hdrop[l - 1]) : at<T>(ws, w.d_hseq[l - 1])) + o;
        a.Kx = Hd;
        a.Wih = at<T>(shadow, so.wih[l]);
        a.bias_gi = at<float>(shadow, so.bias_gi[l]);
      }
      a.h_op = hseq + o;
      a.h32 = at<float>(ws, w.d_carry[l]);
      if (save) {
        a.sv_r = at<T>(ws, w.d_sv[l][0]) + o; a.sv_z = at<T>(ws, w.d_sv[l][1]) + o; a.sv_n = at<T>(ws, w.d_sv[l][2]) + o;
        a.sv_hn = at<T>(ws, w.d_sv[l][3]) + o; a.sv_hprev = at<T>(ws, w.d_sv[l][4]) + o;
      }
      if (drop && l < L - 1) {
        a.h_drop = at<T>(ws, w.d_hdrop[l]) + o;
        a.p_drop = p_drop; a.seed = seed; a.layer = l; a.t = t; a.b_mul = b_mul; a.b_add = b_add;
      }
    }
    for (int i = 0; i < dg.n; ++i)   // algorithmic flop: gates = h_prev.W_hh^T (+ x.W_ih^T when fused)
      pflop += 2.0 * dg.a[i].rows * 3.0 * Hd * (Hd + (dg.a[i].gi_mode == PG_GI_FUSED ? dg.a[i].Kx : 0));
    PG_TRY(launch_gru_fwd_diag<T>(dg, st));
  }
  prof_end(0, pslot, st, sd.T + L - 1, pflop);
  if (Vp == 64 && !getenv("PG_NO_FUSED_CE")) {   // vocab projection + log-softmax + CE + dlogits in one kernel
    // K = 512 is a latency chain of 8 K tiles per block: two of them in flight and eight waves (16 epilogue rows per wave
    // instead of 32) took the kernel from 52 to 36 us; four in flight, or 64-row blocks, measured slower
    using FCfg = TileCfg<T, 128, 64, 8, 4, 2, 1, 2>;
    static_assert(FCfg::SMEM_BYTES >= FCfg::BM * 65 * 4, "fc_ce kernel: LDS too small for the logits tile");
    PG_LAUNCH_BIG_SMEM(fc_ce_kernel<FCfg>, dim3(cdiv(Nd, FCfg::BM)), dim3(FCfg::NTHREADS), FCfg::SMEM_BYTES, st,
                       at<T>(ws, w.d_hseq[L - 1]), Nd, Hd, at<T>(shadow, so.fc_w), at<float>(shadow, so.fc_b), d->V, Vp,
                       at<int>(ws, w.dec_tgt), d->pad, logits_packed ? at<float>(ws, w.d_logits) : (float*)nullptr,   // logits only when the caller takes them
                       save ? at<T>(ws, w.d_dlogits) : (T*)nullptr, scal, dp ? dp->ce_count : 0.0f, recon);   // (its last block finalises the loss)
    PG_CHECK_LAUNCH();
  } else {
    {  // vocab projection
      GemmEpi<T> e;
      e.bias = at<float>(shadow, so.fc_b); e.C32 = at<float>(ws, w.d_logits); e.ldc = Vp;
      PG_TRY(launch_gemm_nt<T>(at<T>(ws, w.d_hseq[L - 1]), Hd, at<T>(shadow, so.fc_w), Hd, (int)Nd, Vp, Hd, e, st));
    }
    if (Vp == 64)
      hipLaunchKernelGGL((ce_kernel<T, 1>), dim3(std::min<int>(cdiv(Nd, 4), 1024)), dim3(256), 0, st, at<float>(ws, w.d_logits),
                         at<int>(ws, w.dec_tgt), Nd, d->V, Vp, d->pad, save ? at<T>(ws, w.d_dlogits) : (T*)nullptr, scal);
    else   // V in (64, 128]: two vocabulary entries per lane
      hipLaunchKernelGGL((ce_kernel<T, 2>), dim3(std::min<int>(cdiv(Nd, 4), 1024)), dim3(256), 0, st, at<float>(ws, w.d_logits),
                         at<int>(ws, w.dec_tgt), Nd, d->V, Vp, d->pad, save ? at<T>(ws, w.d_dlogits) : (T*)nullptr, scal);
    PG_CHECK_LAUNCH();
    hipLaunchKernelGGL(finalize_ce_kernel, dim3(1), dim3(1), 0, st, scal, dp ? dp->ce_count : 0.0f, recon);
    PG_CHECK_LAUNCH();
  }
  if (logits_packed) *logits_packed = at<float>(ws, w.d_logits);
  return PG_OK;
}

extern "C" int pg_decoder_fwd(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes,
                              int B, int Tmax, const int32_t* d_tokens, const int32_t* h_lengths, const float* z,
                              const pg_dp* dp, float p_drop, uint64_t seed, int all_steps, int save_for_backward,
                              float* recon, float** logits_packed, void* stream) {
  PG_TRY(check_dims(d, dtype));
  PG_TRY(check_persist());
  PG_REQUIRE(params && shadow && ws && d_tokens && z, "pg_decoder_fwd: NULL buffer");
  return dtype == PG_BF16     // (the shadow join sits inside, behind the prologue that may share the side stream)
             ? decoder_fwd_t<bf16_t>(d, dtype, params, shadow, ws, ws_bytes, B, Tmax, d_tokens, h_lengths, z, dp, p_drop, seed,
                                     all_steps, save_for_backward, recon, logits_packed, (hipStream_t)stream)
             : decoder_fwd_t<float>(d, dtype, params, shadow, ws, ws_bytes, B, Tmax, d_tokens, h_lengths, z, dp, p_drop, seed,
                                    all_steps, save_for_backward, recon, logits_packed, (hipStream_t)stream);
}

// ===================================================================================================
// decoder backward (BPTT)
// ===================================================================================================
template <typename T>
static int decoder_bwd_t(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes, int B,
                         int Tmax, const int32_t* lens, const pg_dp* dp, float p_drop, uint64_t seed, int all_steps,
                         float g_host, const float* g_dev, float* grads, float* dz, hipStream_t st) {
  const ParamOff po = param_offsets(d);
  const ShadowOff so = shadow_offsets(d, dtype);
  Sched sd;
  PG_TRY(make_sched(lens, B, Tmax, all_steps ? 0 : 1, &sd));
  const long Ne = sum_len(lens, B), Nd = sd.off[sd.T];
  const TrainWs w = train_ws_layout(d, dtype, B, Ne, Ne);
  PG_REQUIRE(ws_bytes >= w.total, "decoder_bwd: workspace too small (%zu < %zu)", ws_bytes, w.total);
  const int Hd = d->Hd, Dz = d->Dz, L = d->L, Vp = vpad(d), V = d->V, E = d->E;
  const int b_mul = dp ? dp->b_mul : 1, b_add = dp ? dp->b_add : 0;
  float* scal = at<float>(ws, w.scalars);
  // scale of the CE gradients = upstream gradient / count: with a host-side upstream gradient (the fused step) the GEMMs
  // take g_host as alpha and 1 / count straight from the forward pass's slot - no one-thread kernel on the critical path
  const bool gs_fold = g_dev == nullptr;
  const float a_ce = gs_fold ? g_host : 1.0f;
  if (!gs_fold)
  hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, st, scal, (int)SC_GS_CE, g_host, g_dev, (int)SC_INV_CNT, 1.0f);
  PG_CHECK_LAUNCH();
  const float* s_ce = scal + (gs_fold ? SC_INV_CNT : SC_GS_CE);
  T* dlog = at<T>(ws, w.d_dlogits);
  const T* top = at<T>(ws, w.d_hseq[L - 1]);
  // (decoder_fc's weight gradient is enqueued with the other weight gradients, behind the recurrence)
  float* dout = at<float>(ws, w.b_dout);
  {
    GemmEpi<T> e;
    e.C32 = dout; e.ldc = Hd; e.alpha = a_ce; e.alpha_dev = s_ce;
    PG_TRY(launch_gemm_nt<T>(dlog, Vp, at<T>(shadow, so.fc_wT), Vp, (int)Nd, Hd, Vp, e, st));
  }
  float* dh0 = at<float>(ws, w.b_dh0);
  const bool drop = p_drop > 0.0f;
  // BPTT as a wavefront: cell (l, t) needs (l, t+1) and (l+1, t); the gradient coming down from layer l+1 is
  // computed inside the cell (dGI_{l+1,t} . W_ih_{l+1}, through layer l's dropout mask), so no per-layer dX GEMM.
  bool persist = false;
  const int sv_lane = sv_layout_of(ws);
  PG_REQUIRE(sv_lane >= 0, "pg_decoder_bwd: no pg_decoder_fwd(save_for_backward = 1) has run on this workspace");
  // the forward pass wrote lane order expecting the persistent BPTT; the per-step BPTT cannot read it (after a give-up or
  // a pg_persist_disable between the two calls the step has to restart from pg_decoder_fwd)
  PG_REQUIRE(!(sv_lane >= 1 && !(std::is_same<T, bf16_t>::value && use_persistent_dec_bwd(Hd, L, B))),
             "pg_decoder_bwd: the forward pass saved its activations for the persistent BPTT kernel, which is now disabled "
             "(PG_ERR_PERSIST or pg_persist_disable since pg_decoder_fwd): restart the step from pg_decoder_fwd");
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (use_persistent_dec_bwd(Hd, L, B)) {   // ONE launch: W_hh^T / W_ih_up^T rows stationary per workgroup, layers pipelined top-down
      DecBwdPersist pa;
      pa.L = L; pa.B = B;
      for (int l = 0; l < L; ++l) {
        pa.WhhT[l] = at<T>(shadow, so.whhT[l]);
        if (l > 0) pa.WihT[l] = at<T>(shadow, so.wihT[l]);
        for (int i = 0; i < 5; ++i) pa.sv[l][i] = at<T>(ws, w.d_sv[l][i]);
        pa.dGI[l] = at<T>(ws, w.b_dGI_l[l]); pa.dGH[l] = at<T>(ws, w.b_dGH_l[l]);
        pa.carry[l] = at<float>(ws, w.b_dhz_l[l][0]); pa.dh0[l] = at<float>(ws, w.b_dhz_l[l][1]);
        pa.dbhh[l] = grads + po.dec_bhh[l]; pa.dbih[l] = grads + po.dec_bih[l];   // bias gradients: not from the GEMMs then
      }
      pa.dout = dout;
      pa.p_drop = p_drop; pa.seed = seed; pa.b_mul = b_mul; pa.b_add = b_add;
      pa.sync = at<unsigned>(ws, w.sync_bwd);
      const int ps = prof_begin(1, st);
      pa.sv_lane = sv_lane;   // what the forward pass of this step recorded (not re-derived: the persist mask may have changed)
      // counters: zeroed by the forward pass of this step unless a BPTT has used them since; mirror: at the end, below
      const bool clean = bwd_sync_record(ws, false);
      PG_TRY(launch_dec_bwd_persist(pa, sd, Hd, dh0, at<T>(ws, w.b_dh0_op), st,
                                    PG_PERSIST_NO_MIRROR | (clean ? PG_PERSIST_SYNC_ZEROED : 0)));
      double pf = 0.0;   // algorithmic flop: dh = dGH_{t+1}.W_hh (+ dGI_{l+1,t}.W_ih_{l+1} below the top layer)
      for (int t = -1; t < sd.T; ++t) {
        const double rin = t + 1 < sd.T ? sched_B(sd, t + 1) : 0, rout = t >= 0 ? sched_B(sd, t) : 0;
        pf += 2.0 * 3.0 * Hd * Hd * (L * rin + (L - 1) * rout);
      }
      prof_end(1, ps, st, 1, pf);
      persist = true;
    }
  }
  const int pslot = persist ? -1 : prof_begin(1, st);
  double pflop = 0.0;
  for (int dgi = 0; dgi < sd.T + L && !persist; ++dgi) {
    GruBwdDiag<T> dg;
    for (int l = L - 1; l >= 0; --l) {
      const int t = sd.T - 1 - (dgi - (L - 1 - l));
      if (t < -1 || t > sd.T - 1) continue;
      GruBwdArgs<T>& a = dg.a[dg.n++];
      T* dGI = at<T>(ws, w.b_dGI_l[l]);
      T* dGH = at<T>(ws, w.b_dGH_l[l]);
      a.H = Hd;
      a.rows_in = (t + 1 < sd.T) ? sched_B(sd, t + 1) : 0;
      a.dGH_in = dGH + (long)sd.off[t + 1 < sd.T ? t + 1 : 0] * 3 * Hd;
      a.WhhT = at<T>(shadow, so.whhT[l]);
      a.dhz_in = at<float>(ws, w.b_dhz_l[l][(t + 1) & 1]);
      if (t >= 0) {
        const long o = (long)sd.off[t] * Hd;
        a.rows_out = sched_B(sd, t);
        if (l == L - 1) {
          a.dout = dout + o;
        } else {
          a.dGI_up = at<T>(ws, w.b_dGI_l[l + 1]) + o * 3;
          a.WihT_up = at<T>(shadow, so.wihT[l + 1]);
          if (drop) { a.p_drop = p_drop; a.seed = seed; a.layer = l; a.t = t; a.b_mul = b_mul; a.b_add = b_add; }
        }
        a.sv_r = at<T>(ws, w.d_sv[l][0]) + o; a.sv_z = at<T>(ws, w.d_sv[l][1]) + o; a.sv_n = at<T>(ws, w.d_sv[l][2]) + o;
        a.sv_hn = at<T>(ws, w.d_sv[l][3]) + o; a.sv_hprev = at<T>(ws, w.d_sv[l][4]) + o;
        a.dGI = dGI + o * 3; a.dGH = dGH + o * 3;
        a.dhz_out = at<float>(ws, w.b_dhz_l[l][t & 1]);
      } else {  // gradient of h0 (shared by all layers); layer L-1 gets there first
        a.rows_out = a.rows_in;
        a.final_step = 1;
        a.dh0 = dh0;
        a.dh0_accumulate = (l != L - 1);
      }
    }
    for (int i = 0; i < dg.n; ++i)   // algorithmic flop: dh = dGH_{t+1}.W_hh (+ dGI_{l+1,t}.W_ih_{l+1} from the layer above)
      pflop += 2.0 * 3.0 * Hd * Hd * (dg.a[i].rows_in + (dg.a[i].dGI_up ? dg.a[i].rows_out : 0));
    PG_TRY(launch_gru_bwd_diag<T>(dg, st));
  }
  prof_end(1, pslot, st, sd.T + L, pflop);
  // Order of the tail.  The large weight-gradient GEMMs need nothing but the recurrence's outputs; the dz chain (row-segment
  // sum, two skinny GEMMs) is ALL the encoder BPTT waits for.  With deferral on the GEMMs go to the side streams and are
  // forked FIRST, right behind the recurrence: as 192 one-per-CU workgroups they leave 64 CUs to the chain's small kernels
  // (with the 128 x 128 engine's 768 resident blocks the chain ran 3x slower beside them and the fork sat behind it).
  static const bool fork_late = getenv("PG_TAIL_FORK_LATE") != nullptr;   // A/B aid: the dz chain in front of the fork
  hipStream_t sg_big = st;
  if (!fork_late) PG_TRY(defer_fork(st, &sg_big));
  const bool fork_early = sg_big != st;
  T* dGI0 = at<T>(ws, w.b_dGI_l[0]);
  float* dzp = at<float>(ws, w.b_dzproj);
  T* dzp_op = at<T>(ws, w.b_dzproj_op);
  T* dh0_op = at<T>(ws, w.b_dh0_op);
  auto dz_chain = [&]() -> int {
    hipLaunchKernelGGL(rowseg_sum_kernel<T>, dim3(cdiv(3 * Hd, 64 * (16 / (int)sizeof(T))), cdiv(B, 4)), dim3(256), 0, st, dGI0,
                       at<int>(ws, w.dec_off), sd.T, 3 * Hd, B, dzp, dzp_op, dz, (long)B * Dz);   // (also zeroes dz for the split-K GEMMs below)
    PG_CHECK_LAUNCH();
    if (!persist) PG_TRY(convert_to<T>(dh0, dh0_op, (long)B * Hd, st));
    // (K split over blocks: 8 output tiles with K = 1536 / 512 were 38 + 32 us of latency chain in front of the encoder BPTT)
    PG_TRY(launch_gemm_nt_splitk<T>(dzp_op, 3 * Hd, at<T>(shadow, so.wih0zT), 3 * Hd, B, Dz, 3 * Hd, dz, Dz, 1.0f, st));
    PG_TRY(launch_gemm_nt_splitk<T>(dh0_op, Hd, at<T>(shadow, so.lat_wT), Hd, B, Dz, Hd, dz, Dz, 1.0f, st));
    return PG_OK;
  };
  if (!fork_early) {
    PG_TRY(dz_chain());
    PG_TRY(defer_fork(st, &sg_big));
  }
  // The five large GEMMs alternate between the two side streams: one of them alone keeps the chip at ~30 % of the MFMA peak
  // (its split-K epilogue is a quarter of its time, and a block's loads are exposed at three blocks per CU); two in flight
  // fill each other's gaps.
  const bool two = sg_big != st && !getenv("PG_TAIL_ONE_STREAM");
  int nbig = 0;
  // (each stream has its own slab buffer for the two-stage split-K reduction of these GEMMs)
  int which = 0;
  // (PG_TN256_PAIR_BLOCKS, A/B aid: workgroups per GEMM for the four that run in pairs - 96 + 96 would leave the dz chain its
  // 64 CUs; the fifth, which runs alone, keeps the default)
  static const int pair_blocks = getenv("PG_TN256_PAIR_BLOCKS") ? atoi(getenv("PG_TN256_PAIR_BLOCKS")) : 0;
  auto big_stream = [&]() {
    g_tn256_blocks_hint = (two && nbig < 4) ? pair_blocks : 0;
    which = (two && (nbig++ & 1)) ? 1 : 0;
    return which ? g_side2 : sg_big;
  };
  const bool slabs = !getenv("PG_TN_ATOMIC");
  hipStream_t sg = sg_big;
  PG_TRY(launch_gemm_tn<T>(dlog, Vp, top, Hd, V, Hd, (int)Nd, grads + po.fc_w, Hd, a_ce, s_ce, sg, grads + po.fc_b));   // decoder_fc
  for (int l = L - 1; l >= 0; --l) {
    T* dGI = at<T>(ws, w.b_dGI_l[l]);
    T* dGH = at<T>(ws, w.b_dGH_l[l]);
    // weight gradients of the layer, batched over all packed rows; the bias gradients (column sums of dGH / dGI) ride along
    {
      hipStream_t sb = big_stream();
      PG_TRY(launch_gemm_tn<T>(dGH, 3 * Hd, at<T>(ws, w.d_sv[l][4]), Hd, 3 * Hd, Hd, (int)Nd, grads + po.dec_whh[l], Hd, 1.0f,
                               nullptr, sb, persist ? nullptr : grads + po.dec_bhh[l], nullptr, slabs ? at<float>(ws, w.b_slab[which]) : nullptr,
                               PG_TN_SLAB_BYTES));
    }
    if (l > 0) {
      const T* X = drop ? at<T>(ws, w.d_hdrop[l - 1]) : at<T>(ws, w.d_hseq[l - 1]);
      hipStream_t sb = big_stream();
      PG_TRY(launch_gemm_tn<T>(dGI, 3 * Hd, X, Hd, 3 * Hd, Hd, (int)Nd, grads + po.dec_wih[l], Hd, 1.0f, nullptr, sb,
                               persist ? nullptr : grads + po.dec_bih[l], nullptr, slabs ? at<float>(ws, w.b_slab[which]) : nullptr,
                               PG_TN_SLAB_BYTES));
    }
  }
  g_tn256_blocks_hint = 0;
  if (fork_early) {
    PG_TRY(dz_chain());
    // the small gradients below read the chain's outputs: the second side stream waits for it
    PG_CHECK_HIP(hipEventRecord(g_ev_fork, st));
    PG_CHECK_HIP(hipStreamWaitEvent(g_side2, g_ev_fork, 0));
  }
  // ---- the small weight gradients feed nothing downstream: on a second side stream when deferring - a string of
  // latency-bound kernels (130 us) that overlaps the large GEMMs instead of queueing behind them - else here in order
  // (measured: they were 170 us of the 430-us chain between the recurrence and the encoder BPTT)
  {
    hipStream_t sg = defer_second(sg_big, st);
    // layer 0 input = [x_emb[token] ; z]: token half through the gather table, z half per sequence
    float* dT = at<float>(ws, w.b_dT);
    // (scratch: the slab buffer of this stream's large GEMMs, which are through in stream order; the rows were grouped by
    // token in the forward pass when it could)
    char* scr = at<char>(ws, w.b_slab[sg == sg_big ? 0 : 1]);
    float* bih0 = persist ? nullptr : grads + po.dec_bih[0];
    if (tok_sorted(ws, 2))
      PG_TRY(launch_tok_sum_sorted<T>(dGI0, 3 * Hd, Nd, 3 * Hd, Vp, tok_sort_at(ws, w.d_ts), reinterpret_cast<float*>(scr),
                                      PG_TN_SLAB_BYTES, dT, bih0, 0, sg));
    else
      PG_TRY(launch_tok_scatter_sum<T>(at<int>(ws, w.dec_tok), dGI0, 3 * Hd, Nd, 3 * Hd, Vp, dT, bih0, scr, PG_TN_SLAB_BYTES, sg));
    // d x_emb[v,e] += sum_g dT[v,g] W_ih0[g,e] ; d W_ih0[g,e] += sum_v dT[v,g] x_emb[v,e]
    PG_TRY(small_gemm(dT, 3 * Hd, 1, params + po.dec_wih[0], E + Dz, 1, V, E, 3 * Hd, grads + po.x_emb, E, d->pad, sg));
    PG_TRY(small_gemm(dT, 1, 3 * Hd, params + po.x_emb, E, 1, 3 * Hd, E, V, grads + po.dec_wih[0], E + Dz, -1, sg));
    PG_TRY(launch_gemm_tn<T>(dzp_op, 3 * Hd, at<T>(ws, w.d_zop), Dz, 3 * Hd, Dz, B, grads + po.dec_wih[0] + E, E + Dz, 1.0f,
                             nullptr, sg));
    // decoder_lat: h0 = W z + b feeds every layer
    PG_TRY(launch_gemm_tn<T>(dh0_op, Hd, at<T>(ws, w.d_zop), Dz, Hd, Dz, B, grads + po.lat_w, Dz, 1.0f, nullptr, sg,
                             grads + po.lat_b));
  }
  // the host mirror of the give-up word, behind the stream of large GEMMs (the shorter of the tail's chains)
  if (persist) PG_TRY(persist_fail_mirror(sg_big));
  PG_TRY(defer_done(sg_big, st));
  return PG_OK;
}

extern "C" int pg_decoder_bwd(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes,
                              int B, int Tmax, const int32_t* h_lengths, const pg_dp* dp, float p_drop, uint64_t seed,
                              int all_steps, float g_host, const float* g_dev, float* grads, float* dz, void* stream) {
  PG_TRY(check_dims(d, dtype));
  PG_TRY(check_persist());
  PG_REQUIRE(params && shadow && ws && grads && dz, "pg_decoder_bwd: NULL buffer");
  PG_TRY(shadow_join((hipStream_t)stream));
  return dtype == PG_BF16 ? decoder_bwd_t<bf16_t>(d, dtype, params, shadow, ws, ws_bytes, B, Tmax, h_lengths, dp, p_drop, seed,
                                                  all_steps, g_host, g_dev, grads, dz, (hipStream_t)stream)
                          : decoder_bwd_t<float>(d, dtype, params, shadow, ws, ws_bytes, B, Tmax, h_lengths, dp, p_drop, seed,
                                                 all_steps, g_host, g_dev, grads, dz, (hipStream_t)stream);
}

// ===================================================================================================
// encoder backward (BPTT)
// ===================================================================================================
template <typename T>
static int encoder_bwd_t(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes, int B,
                         int Tmax, const int32_t* lens, const pg_dp* dp, const float* dz, float g_host, const float* g_dev,
                         float* grads, hipStream_t st) {
  const ParamOff po = param_offsets(d);
  const ShadowOff so = shadow_offsets(d, dtype);
  Sched se;
  PG_TRY(make_sched(lens, B, Tmax, 0, &se));
  const long Ne = se.off[se.T];
  const TrainWs w = train_ws_layout(d, dtype, B, Ne, Ne);
  PG_REQUIRE(ws_bytes >= w.total, "encoder_bwd: workspace too small (%zu < %zu)", ws_bytes, w.total);
  const int Hq = d->Hq, Dz = d->Dz, Vp = vpad(d), V = d->V, E = d->E;
  float* scal = at<float>(ws, w.scalars);
  const float inv_B = (dp && dp->kl_inv_batch > 0.0f) ? dp->kl_inv_batch : 1.0f / (float)B;
  float* dq = at<float>(ws, w.b_dq);
  T* dq_op = at<T>(ws, w.b_dq_op);
  hipLaunchKernelGGL(reparam_bwd_kernel<T>, dim3(cdiv((long)B * Dz, 256)), dim3(256), 0, st, at<float>(ws, w.e_mulv),
                     at<float>(ws, w.e_eps), dz, g_host * inv_B, g_dev, B, Dz, dq, dq_op);
  PG_CHECK_LAUNCH();
  const T* hfin = at<T>(ws, w.e_hfin_op);
  // The q heads' weight gradients feed nothing downstream, and in front of the recurrence they sit in the window in which the
  // decoder's 256 x 256 weight-gradient workgroups hold every CU's LDS (the first of them waits ~50 us for a CU).  Moving
  // them behind the recurrence (PG_QGRAD_LATE=1, end of the embedding tail's stream) only hands the wait to the next
  // LDS-heavy kernel of the chain, and the recurrence then ends while the side streams' last GEMMs still run: +0.02 ms.
  static const bool qgrad_early = getenv("PG_QGRAD_LATE") == nullptr;
  auto q_grads = [&](hipStream_t sq) -> int {
    PG_TRY(launch_gemm_tn<T>(dq_op, 2 * Dz, hfin, Hq, Dz, Hq, B, grads + po.qmu_w, Hq, 1.0f, nullptr, sq, grads + po.qmu_b));
    PG_TRY(launch_gemm_tn<T>(dq_op + Dz, 2 * Dz, hfin, Hq, Dz, Hq, B, grads + po.qlv_w, Hq, 1.0f, nullptr, sq, grads + po.qlv_b));
    return PG_OK;
  };
  if (qgrad_early) PG_TRY(q_grads(st));
  float* dhfin = at<float>(ws, w.b_dhfin);
  {
    GemmEpi<T> e;
    e.C32 = dhfin; e.ldc = Hq;
    PG_TRY(launch_gemm_nt<T>(dq_op, 2 * Dz, at<T>(shadow, so.wqT), 2 * Dz, B, Hq, 2 * Dz, e, st));
  }
  T* dGI = at<T>(ws, w.b_dGI);
  T* dGH = at<T>(ws, w.b_dGH);
  bool persist = false;
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (use_persistent(Hq)) {
      EncBwdPersist pa;
      pa.WhhT = at<T>(shadow, so.enc_whhT); pa.dhfin = dhfin;
      pa.sv_r = at<T>(ws, w.e_sv[0]); pa.sv_z = at<T>(ws, w.e_sv[1]); pa.sv_n = at<T>(ws, w.e_sv[2]);
      pa.sv_hn = at<T>(ws, w.e_sv[3]); pa.sv_hprev = at<T>(ws, w.e_sv[4]);
      pa.dGI = dGI; pa.dGH = dGH; pa.B = B;
      PG_TRY(launch_enc_bwd_persist(pa, se, Hq, st));
      persist = true;
    }
  }
  for (int t = se.T - 1; t >= 0 && !persist; --t) {
    GruBwdArgs<T> a;
    a.H = Hq;
    a.rows_out = sched_B(se, t);
    a.rows_in = (t + 1 < se.T) ? sched_B(se, t + 1) : 0;
    a.dGH_in = dGH + (long)se.off[t + 1 < se.T ? t + 1 : 0] * 3 * Hq;
    a.WhhT = at<T>(shadow, so.enc_whhT);
    a.dhz_in = at<float>(ws, w.b_dhz[(t + 1) & 1]);
    a.inject = dhfin;
    const long o = (long)se.off[t] * Hq;
    a.sv_r = at<T>(ws, w.e_sv[0]) + o; a.sv_z = at<T>(ws, w.e_sv[1]) + o; a.sv_n = at<T>(ws, w.e_sv[2]) + o;
    a.sv_hn = at<T>(ws, w.e_sv[3]) + o; a.sv_hprev = at<T>(ws, w.e_sv[4]) + o;
    a.dGI = dGI + o * 3; a.dGH = dGH + o * 3;
    a.dhz_out = at<float>(ws, w.b_dhz[t & 1]);
    PG_TRY(launch_gru_bwd<T>(a, st));
  }
  // The embedding-table tail (one-hot^T . dGI, then two small GEMMs) feeds nothing downstream: with deferral on it goes to
  // the second side stream, beside the W_hh gradient below, instead of queueing behind it on the caller's stream.  Own
  // one-hot / table-gradient scratch and atomic accumulation into x_emb's gradient: the decoder's deferred tail may still
  // be running there.
  // (PG_ENC_TAIL_INORDER=1, A/B aid: everything on the caller's stream, the table gradient in front of the W_hh gradient -
  // measured 0.02 ms per step slower.)
  static const bool tail_side = getenv("PG_ENC_TAIL_INORDER") == nullptr;
  hipStream_t se2 = st;
  if (tail_side && g_defer && g_defer_pending && g_side2) {
    PG_CHECK_HIP(hipEventRecord(g_ev_fork, st));
    PG_CHECK_HIP(hipStreamWaitEvent(g_side2, g_ev_fork, 0));
    se2 = g_side2;
  }
  float* dT = at<float>(ws, w.e_dT);
  // (rows grouped by token in the forward pass, or here; partial sums in the encoder's own slab - the decoder's deferred GEMMs
  // may still be using theirs.  More rows than one sort takes: several passes through the second side stream's slab buffer,
  // behind the join with that stream.)
  bool sorted = tok_sorted(ws, 1);
  if (!sorted && Ne <= tok_sort_max_rows(Vp)) {
    PG_TRY(launch_tok_sort(at<int>(ws, w.enc_tok), Ne, Vp, tok_sort_at(ws, w.e_ts), se2));
    sorted = true;
  }
  if (sorted) {
    PG_TRY(launch_tok_sum_sorted<T>(dGI, 3 * Hq, Ne, 3 * Hq, Vp, tok_sort_at(ws, w.e_ts), at<float>(ws, w.e_slab),
                                    (size_t)tok_sort_chunks(Ne, Vp) * 3 * Hq * sizeof(float), dT, grads + po.enc_bih, 0, se2));
  } else {
    if (se2 == st) PG_TRY(defer_join(st, false));
    PG_TRY(launch_tok_scatter_sum<T>(at<int>(ws, w.enc_tok), dGI, 3 * Hq, Ne, 3 * Hq, Vp, dT, grads + po.enc_bih,
                                     at<char>(ws, w.b_slab[1]), PG_TN_SLAB_BYTES, se2));
  }
  PG_TRY(small_gemm(dT, 3 * Hq, 1, params + po.enc_wih, E, 1, V, E, 3 * Hq, grads + po.x_emb, E, d->pad, se2));
  PG_TRY(small_gemm(dT, 1, 3 * Hq, params + po.x_emb, E, 1, 3 * Hq, E, V, grads + po.enc_wih, E, -1, se2));
  if (!qgrad_early) PG_TRY(q_grads(se2));    // (behind the embedding tail: the shorter of the two chains that end the step)
  if (se2 != st) PG_TRY(defer_rerecord());   // the joins below and later wait for this newer record
  PG_TRY(launch_gemm_tn<T>(dGH, 3 * Hq, at<T>(ws, w.e_sv[4]), Hq, 3 * Hq, Hq, (int)Ne, grads + po.enc_whh, Hq, 1.0f, nullptr, st,
                           grads + po.enc_bhh));
  PG_TRY(defer_join(st, false));   // the decoder's deferred weight gradients: complete in stream order from here on
  return PG_OK;
}

extern "C" int pg_encoder_bwd(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes,
                              int B, int Tmax, const int32_t* h_lengths, const pg_dp* dp, const float* dz, float g_host,
                              const float* g_dev, float* grads, void* stream) {
  PG_TRY(check_dims(d, dtype));
  PG_TRY(check_persist());
  PG_REQUIRE(params && shadow && ws && grads, "pg_encoder_bwd: NULL buffer");
  PG_REQUIRE(!enc_generic(d), "pg_encoder_bwd serves the one-layer one-direction encoder; Lq=%d qbidir=%d goes to pg_encoder_gen_bwd",
             d->Lq, d->qbidir);
  PG_TRY(shadow_join((hipStream_t)stream));   // (the forward pass's token sort may have run on the side stream)
  return dtype == PG_BF16 ? encoder_bwd_t<bf16_t>(d, dtype, params, shadow, ws, ws_bytes, B, Tmax, h_lengths, dp, dz, g_host, g_dev,
                                                  grads, (hipStream_t)stream)
                          : encoder_bwd_t<float>(d, dtype, params, shadow, ws, ws_bytes, B, Tmax, h_lengths, dp, dz, g_host, g_dev,
                                                 grads, (hipStream_t)stream);
}

// ===================================================================================================
// optimiser
// ===================================================================================================
static int clip_adam_impl(long n, float* params, const float* grads, float* zero_g, float* exp_avg, float* exp_avg_sq, int step,
                          float lr, float beta1, float beta2, float eps, float max_norm, float* scratch, float* gnorm,
                          hipStream_t st, const char* who) {
  PG_REQUIRE(n > 0 && params && grads && exp_avg && exp_avg_sq && scratch, "%s: NULL buffer", who);
  PG_REQUIRE(step >= 1, "%s: step must be >= 1", who);
  PG_TRY(check_persist());
  PG_TRY(defer_join(st, true));
  hipLaunchKernelGGL(sumsq_kernel, dim3(PG_SUMSQ_BLOCKS), dim3(256), 0, st, grads, n, scratch);
  PG_CHECK_LAUNCH();
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, scratch,
                     (int)PG_SUMSQ_BLOCKS, max_norm, (float)(lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps, gnorm,
                     persist_fail_dev_or_null(), zero_g);
  PG_CHECK_LAUNCH();
  return PG_OK;
}
extern "C" int pg_clip_adam(long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int step, float lr,
                            float beta1, float beta2, float eps, float max_norm, float* scratch, float* gnorm, void* stream) {
  return clip_adam_impl(n, params, grads, nullptr, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, max_norm, scratch, gnorm,
                        (hipStream_t)stream, "pg_clip_adam");
}
extern "C" int pg_clip_adam_zero(long n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, int step, float lr,
                                 float beta1, float beta2, float eps, float max_norm, float* scratch, float* gnorm, void* stream) {
  return clip_adam_impl(n, params, grads, grads, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, max_norm, scratch, gnorm,
                        (hipStream_t)stream, "pg_clip_adam_zero");
}

// ---- parity hook: the inter-layer dropout masks of pg_decoder_fwd / pg_decoder_bwd, computed on the HOST by the very
// function the kernels call (dropout_scale4, pg_common.h), so that a test can hand the same masks to a CPU oracle.
// out (host) [n_masks][B][T][H]: mask of layer l's output (l = 0 .. n_masks-1) at decoder step t of local row b, already
// scaled by 1/(1-p).
extern "C" int pg_dropout_masks_host(uint64_t seed, float p_drop, int n_masks, int B, int T, int H, int b_mul, int b_add,
                                     float* out) {
  PG_REQUIRE(out && n_masks >= 0 && B >= 1 && T >= 1 && H >= 4 && H % 4 == 0, "pg_dropout_masks_host: bad arguments");
  PG_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "pg_dropout_masks_host: dropout %f outside [0,1)", p_drop);
  const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  for (int l = 0; l < n_masks; ++l)
    for (int b = 0; b < B; ++b)
      for (int t = 0; t < T; ++t) {
        float* o = out + (((size_t)l * B + b) * T + t) * H;
        for (int c4 = 0; c4 < H / 4; ++c4) {
          float m[4] = {1.0f, 1.0f, 1.0f, 1.0f};
          if (p_drop > 0.0f) dropout_scale4(seed, l, t, b * b_mul + b_add, c4, p_drop, inv_keep, m);
          o[4 * c4] = m[0]; o[4 * c4 + 1] = m[1]; o[4 * c4 + 2] = m[2]; o[4 * c4 + 3] = m[3];
        }
      }
  return PG_OK;
}

// ---- the one entry point that allocates nothing (include/polygon_hip.h: pg_clip_adam_static) -------------------------
// pg_clip_adam with the partial sums of squares (the only device memory of the optimiser step beyond its inputs and
// outputs) in a library-owned array instead of a caller workspace.  Every GPU the library runs on holds its own copy of
// the array at its own address: the address is taken once per device, with that device current, and only that pointer
// is ever handed to kernels (the array's name is a host-side stand-in, not a device address).
static __device__ __align__(256) float scratch[PG_SUMSQ_BLOCKS];
constexpr int PG_STATIC_MAX_DEVICES = 64;
static float* g_scratch_ptr[PG_STATIC_MAX_DEVICES] = {};

extern "C" int pg_clip_adam_static(long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int step,
                                   float lr, float beta1, float beta2, float eps, float max_norm, float* gnorm, void* stream) {
  PG_REQUIRE(n > 0 && n <= PG_CLIP_ADAM_STATIC_MAX_N,
             "pg_clip_adam_static: n = %ld outside [1, %ld] (the library-owned partial-sum array is sized for that many values)", n,
             (long)PG_CLIP_ADAM_STATIC_MAX_N);
  int dev = -1;
  PG_CHECK_HIP(hipGetDevice(&dev));
  PG_REQUIRE(dev >= 0 && dev < PG_STATIC_MAX_DEVICES, "pg_clip_adam_static: device ordinal %d not supported", dev);
  if (!g_scratch_ptr[dev]) {
    float* p = nullptr;
    PG_CHECK_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
    PG_REQUIRE(p != nullptr, "pg_clip_adam_static: no device address for the scratch array on device %d", dev);
    g_scratch_ptr[dev] = p;
  }
  // every partial is rewritten by sumsq_kernel on each call (nothing has to start at zero), and adam_kernel reads them in
  // stream order: calls on ONE stream are safe back to back, calls overlapping on one GPU (two streams) race on the array
  return pg_clip_adam(n, params, grads, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, max_norm, g_scratch_ptr[dev], gnorm,
                      stream);
}
