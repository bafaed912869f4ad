/* polygon_hip.h -- C ABI of libpolygon_hip.so: the MI355X (gfx950) hot path of POLYGON's SMILES VAE.
 *
 * The reference (bpmunson/polygon) has no FFI boundary: its hot path is the Python class
 * polygon/vae/vae_model.py:VAE whose arithmetic is delegated to stock torch modules.  Each entry
 * point below replaces the torch call sites named in its comment (file:line relative to the
 * reference root); the Python mirror of the reference API that binds them is polygon_amd/vae/.
 *
 * Conventions: every function returns 0 or a negative PG_ERR_* code (-1 argument, -2 HIP, -3 unsupported,
 * -4 workspace, -5 a persistent kernel gave up: see pg_check_failure); pg_last_error() returns the
 * calling thread's last message.  Device pointers are raw HIP pointers; work is enqueued on the
 * caller's stream (hipStream_t passed as void*) with no host synchronisation, and every per-call
 * buffer (workspaces sized by pg_*_workspace_bytes) is the caller's.  No C++ exception crosses the
 * boundary.
 *
 * Library-owned state (process-wide, NOT thread-safe: one process per GPU, one calling thread at a
 * time is the model): (1) a 64-byte device word + 64-byte pinned host mirror "a persistent kernel
 * gave up", allocated on the first persistent launch and never freed; (2) the TWO side streams and five
 * events (fork, two joins, shadow rebuild, tokens packed) of pg_defer_weight_grads, created on first use; (3) up to
 * 2 x 512 HIP event pairs while pg_profile_enable is on; (4) the switches set by pg_persist_disable /
 * pg_defer_weight_grads / pg_profile_enable / pg_persist_spin_limit / pg_entry_checks (PG_NO_PERSIST is
 * read once at first use) - process-global and NOT thread-safe: the Python layer toggles
 * pg_defer_weight_grads around every training step, so two threads must not drive one process's
 * library at once; (5) the 2-KiB static device array of pg_clip_adam_static; (6) a 64-entry host table
 * "layout of the saved activations per workspace" (pg_decoder_fwd -> pg_decoder_bwd) and two 8-entry ones
 * ("BPTT counters still cleared", "rows grouped by token") written by the forward calls for the backward calls.  Nothing else
 * persists between calls.
 *
 * Parameters are ONE flat fp32 buffer in the order of model.vae.parameters()
 * (vae_model.py:101-115, = optimiser order vae_trainer.py:99-100):
 *   x_emb.weight[V,E]; encoder_rnn {weight_ih[3Hq,E], weight_hh[3Hq,Hq], bias_ih, bias_hh};
 *   q_mu {weight[Dz,Hq], bias}; q_logvar {weight, bias}; decoder_lat {weight[Hd,Dz], bias};
 *   decoder_rnn layer l {weight_ih[3Hd, l?Hd:E+Dz], weight_hh[3Hd,Hd], bias_ih, bias_hh} l=0..L-1;
 *   decoder_fc {weight[V,Hd], bias}.   Gradients use the same layout and are ACCUMULATED (+=).
 *
 * Batches: tokens int32 [B,Tmax] row-major on the device, padded with pad; lengths int32 [B] on the
 * HOST (bos..eos inclusive), sorted descending exactly as the reference collate does
 * (vae_model.py:142-147); unsorted lengths are rejected like torch's pack_sequence.
 */
#ifndef POLYGON_HIP_H
#define POLYGON_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PG_API __attribute__((visibility("default")))
#define PG_VERSION 1
#define PG_F32 0  /* f32 operands on v_mfma_f32_32x32x2_f32 (exact f32) */
#define PG_BF16 1 /* bf16 operands on v_mfma_f32_32x32x16_bf16, f32 accumulate, f32 state */
#define PG_MAX_LAYERS 8
#define PG_MAX_T 511  /* longest sequence (bos..eos) the packed scheduler accepts */
#define PG_MAX_ENC_LAYERS 4 /* q_n_layers limit of the generic encoder */
#define PG_MAX_V 128  /* vocabulary limit of the softmax / cross-entropy / draw kernels */
/* Padded vocabulary width Vp of the logits, probabilities and gather tables: 64 for V <= 64, else 128.  V <= 64 (the
 * reference's alphabet has 51 symbols) is the tuned case: fused projection + cross entropy, one-launch sampler; above it
 * the projection is a plain GEMM, the cross entropy and the draw take two vocabulary entries per lane, and pg_sample
 * runs the per-step kernels. */
#define PG_VPAD(V) ((V) <= 64 ? 64 : 128)

typedef struct pg_dims {
  int V, E, Hq, Dz, Hd, L; /* vocab, embedding dim, encoder hidden, latent, decoder hidden, decoder layers */
  int pad, bos, eos, unk;
  /* encoder shape (vae_model.py:63-81: q_n_layers, q_bidir).  0 / 0 = the reference's defaults (one layer, one
   * direction), so a zero-initialised tail keeps its meaning.  Any other shape changes the flat parameter layout
   * (pg_param_count: nn.GRU's order weight_ih, weight_hh, bias_ih, bias_hh per layer and direction, q_mu / q_logvar
   * weights Dz x Hq*dirs) and is served by pg_encoder_gen_fwd / pg_encoder_gen_bwd; pg_encoder_fwd / _bwd refuse it. */
  int Lq, qbidir;
} pg_dims;

/* data-parallel context of one call: rank r of G holds rows r, r+G, ... of the
 * globally sorted batch.  kl_inv_batch = 1/B_global; ce_count = global number of non-pad targets
 * (<= 0: use the locally counted value); b_mul/b_add map a local row to its global row
 * (b*b_mul + b_add) so the dropout mask (a counter hash of seed, layer, t, global row, column) is independent of G. */
typedef struct pg_dp {
  float kl_inv_batch;
  float ce_count;
  int b_mul, b_add;
} pg_dp;

PG_API int pg_version(void);
PG_API const char* pg_last_error(void);
PG_API long pg_param_count(const pg_dims* d);

/* Derived ("shadow") weights: operand-dtype copies, transposes, the embedding x W_ih gather tables
 * that replace nn.Embedding + the input half of the GRU (vae_model.py:224,253-256).  Rebuild after
 * every parameter update. */
PG_API size_t pg_shadow_bytes(const pg_dims* d, int dtype);
PG_API int pg_prepare_weights(const pg_dims* d, int dtype, const float* params, void* shadow, void* stream);

PG_API size_t pg_train_workspace_bytes(const pg_dims* d, int dtype, int B, long n_tokens);

/* VAE.forward_encoder (vae_model.py:216-239): embedding gather + GRU encoder + q_mu/q_logvar +
 * reparameterisation + KL.  eps [B,Dz] replaces torch.randn_like (NULL: z = mu).  Outputs (device):
 * z [B,Dz], mu/logvar [B,Dz] (nullable), kl scalar (nullable).  save_for_backward keeps the BPTT
 * state in the workspace for pg_encoder_bwd. */
PG_API int pg_encoder_fwd(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes,
                          int B, int Tmax, const int32_t* d_tokens, const int32_t* h_lengths, const float* eps,
                          const pg_dp* dp, int save_for_backward, float* z, float* mu, float* logvar, float* kl,
                          void* stream);

/* VAE.forward_decoder (vae_model.py:241-275): z-conditioned L-layer GRU decoder (teacher forced),
 * vocab projection, log-softmax, cross entropy with ignore_index=pad.  p_drop/seed: inter-layer
 * dropout of nn.GRU in train mode (vae_model.py:90).  all_steps=1 also runs each row's last (eos
 * input) step, which only matters when the logits are read back (return_y).  logits_packed (nullable)
 * receives the device address of the packed [sum(steps), PG_VPAD(V)] fp32 logits inside the workspace. */
PG_API int pg_decoder_fwd(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes,
                          int B, int Tmax, const int32_t* d_tokens, const int32_t* h_lengths, const float* z,
                          const pg_dp* dp, float p_drop, uint64_t seed, int all_steps, int save_for_backward,
                          float* recon, float** logits_packed, void* stream);

/* Hand-written BPTT of the two calls above (what loss.backward() does at vae_trainer.py:134).
 * Upstream gradient of recon / kl = g_host * (*g_dev if g_dev else 1).  pg_decoder_bwd writes
 * dz [B,Dz]; pg_encoder_bwd consumes it (NULL: no reconstruction path).  Both must see the same
 * batch arguments and workspace as their forward call. */
PG_API int pg_decoder_bwd(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes,
                          int B, int Tmax, const int32_t* h_lengths, const pg_dp* dp, float p_drop, uint64_t seed,
                          int all_steps, float g_host, const float* g_dev, float* grads, float* dz, void* stream);
PG_API int pg_encoder_bwd(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes,
                          int B, int Tmax, const int32_t* h_lengths, const pg_dp* dp, const float* dz, float g_host,
                          const float* g_dev, float* grads, void* stream);

/* VAE.forward_encoder for ANY encoder shape of vae_model.py:63-81 (pg_dims.Lq layers, pg_dims.qbidir directions) and its
 * BPTT: nn.GRU(num_layers=q_n_layers, bidirectional=q_bidir, dropout=q_dropout) over the packed batch, the last layer's
 * final state(s) concatenated forward|reverse (vae_model.py:228-229) into q_mu / q_logvar, reparameterisation and KL as
 * pg_encoder_fwd.  Runs on the per-step GRU kernels (one launch per step, layer and direction; the reverse direction
 * walks t = T-1 .. 0 and starts each sequence at its own last token, which is what a packed bidirectional GRU does) and
 * the batched GEMM kernels for the input projections; it keeps its own operand copies of the encoder weights in ITS
 * workspace (pg_encoder_gen_workspace_bytes, separate from the training workspace), rebuilt by every forward call, so it
 * reads nothing of the shadow buffer.  p_drop / seed: nn.GRU's inter-layer dropout (q_dropout, only when Lq > 1), the
 * same counter hash as the decoder's (pg_dropout_masks_host with H = Hq*dirs gives the masks of layers 0 .. Lq-2).
 * Same argument meaning, outputs and error behaviour as pg_encoder_fwd / pg_encoder_bwd otherwise; the default shape is
 * accepted too (a slower path to the same numbers, used by the tests). */
PG_API size_t pg_encoder_gen_workspace_bytes(const pg_dims* d, int dtype, int B, long n_tokens);
PG_API int pg_encoder_gen_fwd(const pg_dims* d, int dtype, const float* params, void* ws, size_t ws_bytes, int B, int Tmax,
                              const int32_t* d_tokens, const int32_t* h_lengths, const float* eps, const pg_dp* dp,
                              float p_drop, uint64_t seed, int save_for_backward, float* z, float* mu, float* logvar,
                              float* kl, void* stream);
PG_API int pg_encoder_gen_bwd(const pg_dims* d, int dtype, const float* params, void* ws, size_t ws_bytes, int B, int Tmax,
                              const int32_t* h_lengths, const pg_dp* dp, float p_drop, uint64_t seed, const float* dz,
                              float g_host, const float* g_dev, float* grads, void* stream);

/* clip_grad_norm_(params, max_norm) + Adam.step() (vae_trainer.py:136-138) over the flat buffers.
 * scratch: >= 4 KiB device floats.  step = 1-based Adam step count.  max_norm <= 0 disables the
 * clip.  gnorm (nullable) receives the pre-clip global L2 norm. */
PG_API int pg_clip_adam(long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int step, float lr,
                        float beta1, float beta2, float eps, float max_norm, float* scratch, float* gnorm, void* stream);

/* pg_clip_adam that leaves `grads` ZEROED (no counterpart in the reference, whose loop calls optimizer.zero_grad() before
 * the next backward, vae_trainer.py:133): the Adam kernel writes the zeros behind its read of each value - also when it
 * skips the update because a persistent kernel gave up - so the next step's gradient accumulation (the entry points above
 * add into `grads`) starts from a clean buffer without a separate fill on the step's critical path. */
PG_API int pg_clip_adam_zero(long n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, int step, float lr,
                             float beta1, float beta2, float eps, float max_norm, float* scratch, float* gnorm, void* stream);

/* pg_clip_adam for callers that cannot allocate or carry a workspace (code under HIP graph capture, tight serving /
 * fine-tuning loops, harnesses that forbid hipMalloc).  NO counterpart in the reference.  Same arithmetic and kernels as
 * pg_clip_adam; the partial sums of squares live in a library-owned static device array (PG_SUMSQ partials, 2 KiB per GPU,
 * held until the process ends) instead of `scratch`.  Allocates nothing (no hipMalloc / hipFree / allocating routine).
 * n > PG_CLIP_ADAM_STATIC_MAX_N is rejected with PG_ERR_ARG.  All calls on one GPU share that GPU's array: calls on one
 * stream may follow each other freely, calls that OVERLAP on one GPU (different streams / threads) race. */
#define PG_CLIP_ADAM_STATIC_MAX_N (1L << 28)
PG_API int pg_clip_adam_static(long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int step,
                               float lr, float beta1, float beta2, float eps, float max_norm, float* gnorm, void* stream);

/* VAE.sample (vae_model.py:300-358): max_len-1 fused decode steps (3 GRU cells + vocab projection +
 * softmax(y/temp) + inverse-CDF draw or argmax + eos bookkeeping).  The reference decodes every row to max_len and
 * discards what follows a row's eos (vae_model.py:342-351); here a row LEAVES the decode at its eos (same tokens: its
 * later draws are never recorded; PG_SAMPLE_NO_COMPACT=1 runs every row to max_len, the round-1 behaviour).
 * uniforms [max_len-1, n] (device) feed the draw of step i at uniforms[(i-1)*n + row]; NULL: Philox
 * keyed by (seed, row_offset + row, step).  tokens uint8 [n,max_len], end_pads int32 [n].
 * bf16, default decoder (Hd 512, 3 layers), a device with >= 256 CUs: the whole loop runs as ONE
 * group-persistent launch per 16384 rows (pg_persist_sample.hip) that needs all 256 workgroups resident, so nothing else should be
 * running on the device during the call; other cases launch 3 cells + 1 head kernel per step.  Both draw the same
 * tokens from the same inputs. */
PG_API size_t pg_sample_workspace_bytes(const pg_dims* d, int dtype, int n);
PG_API int pg_sample(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes, int n,
                     int max_len, const float* z, const float* uniforms, uint64_t seed, long row_offset, float temp,
                     int multinomial, uint8_t* tokens, int32_t* end_pads, void* stream);

/* Parity hook (no reference counterpart): pg_sample with teacher forcing and a probability dump, so that the sampling
 * kernels themselves (per-step and one-launch alike) can be compared with a CPU oracle step by step, which free-running
 * bf16 decoding cannot be (one differently rounded draw diverges the row).  forced (nullable) uint8 [n, max_len]: the token
 * FED to step i+1 is forced[row*max_len + i] instead of the drawn one (the drawn token is still what tokens/end_pads
 * record).  probs_out (nullable, multinomial only) float [max_len-1, n, Vp], Vp = PG_VPAD(V): softmax(y/temp) of step i at
 * probs_out[((i-1)*n + row)*Vp + v], entries >= V zero.  Both NULL: exactly pg_sample. */
PG_API int pg_sample_forced(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes,
                            int n, int max_len, const float* z, const float* uniforms, uint64_t seed, long row_offset,
                            float temp, int multinomial, uint8_t* tokens, int32_t* end_pads, const uint8_t* forced,
                            float* probs_out, void* stream);

/* Parity hook (no reference counterpart; HOST function, touches no GPU): the inter-layer dropout masks that
 * pg_decoder_fwd / pg_decoder_bwd apply for (seed, p_drop, pg_dp.b_mul/b_add), computed by the same inline function the
 * kernels call.  out (host memory) float [n_masks, B, T, H]: mask of layer l's output at decoder step t of local row b,
 * already scaled by 1/(1-p).  torch's own nn.GRU dropout stream (vae_model.py:90) cannot be reproduced from outside, so
 * the dropout path is pinned by feeding THESE masks to the CPU oracle. */
PG_API int pg_dropout_masks_host(uint64_t seed, float p_drop, int n_masks, int B, int T, int H, int b_mul, int b_add,
                                 float* out);

/* Kernel-level test hooks (no reference counterpart; they expose the MFMA tile engines that every
 * entry point above is built from, so tests can check them in isolation).
 * pg_gemm_nt: C[M,N] = alpha * A[M,K] . B[N,K]^T (+bias[N]) (+C32 if accumulate) -> C32 (fp32) and/or Ct (dtype).
 * pg_gemm_tn: C[M,N] += alpha * A[R,M]^T . B[R,N]  (split over R across workgroups; this entry point, which lends no slab
 * buffer, combines the splits with fp32 atomics into C; pg_gemm_tn_ws below can use slabs + a reduce kernel). */
PG_API int pg_gemm_nt(int dtype, const void* A, long lda, const void* B, long ldb, int M, int N, int K, const float* bias,
                      float* C32, void* Ct, long ldc, int accumulate, float alpha, void* stream);
PG_API int pg_gemm_tn(int dtype, const void* A, long lda, const void* B, long ldb, int M, int N, int R, float* C, long ldc,
                      float alpha, void* stream);
/* pg_gemm_tn_ws: the same with what the training step passes: colsum_a[M] / colsum_b[N] (nullable) += alpha * column sums
 * of A / B (bias gradients), and a slab buffer (nullable; PG_TN_SLAB_BYTES of pg_gemm.h covers every shape) that switches
 * large outputs to the two-stage split-K reduction and, in bf16 with M and N multiples of 256, to the 256 x 256 LDS-DMA
 * kernel. */
PG_API int pg_gemm_tn_ws(int dtype, const void* A, long lda, const void* B, long ldb, int M, int N, int R, float* C, long ldc,
                         float alpha, float* colsum_a, float* colsum_b, float* slab, size_t slab_bytes, void* stream);

/* pg_tok_scatter_sum: the gradient of a gather table (the embedding folded into the layer-0 input projection): dT[v, 0..C) =
 * sum of G[r, 0..C) over the rows r with tok[r] = v, v < Vp <= 128 (rows with a token outside [0, Vp) are left out), and
 * colsum[c] (nullable) += the sum over all those rows.  G: [N, ldg] in dtype, 16-byte aligned rows, C a multiple of
 * 16 / sizeof(dtype).  scratch: 256-byte aligned; pg_tok_scatter_scratch_bytes(N, C, Vp) bytes take the rows in one pass, less
 * (down to the bytes of a 128-row pass) takes them in several.  dT is overwritten, not accumulated into. */
PG_API size_t pg_tok_scatter_scratch_bytes(long N, int C, int Vp);
PG_API int pg_tok_scatter_sum(int dtype, const int32_t* tok, const void* G, long ldg, long N, int C, int Vp, float* dT,
                              float* colsum, void* scratch, size_t scratch_bytes, void* stream);

/* pg_bench_gru: launches one recurrent step kernel (which = 0 forward cell, 1 backward cell) `repeat` times on
 * scratch operands carved from ws (>= 64*rows*H bytes + 16*H*H bytes): used by bench.py to time the dominant kernel
 * with HIP events, and by the kernel-tuning scripts under tools/. */
PG_API int pg_bench_gru(int dtype, int which, int rows, int H, void* ws, size_t ws_bytes, int repeat, void* stream);

/* Measurement hook (no reference counterpart): while enabled, pg_decoder_fwd / pg_decoder_bwd bracket their
 * recurrence - ONE launch of dec_fwd_persist_kernel / dec_bwd_persist_kernel where the persistent kernels cover the
 * shape (bf16, Hd = 512), otherwise the loop of gru_fwd2_diag_kernel / gru_bwd2_diag_kernel launches, one per
 * anti-diagonal of the (layer, t) grid - with a HIP event pair recorded on the caller's stream and count its launches
 * and algorithmic flop.  pg_profile_read waits for the recorded events, returns the sums since the last read for
 * which = 0 (forward recurrence) or 1 (BPTT recurrence) and clears that side.  At most 512 segments are kept per side
 * between reads; later ones are dropped (not counted). */
PG_API int pg_profile_enable(int on);
PG_API int pg_profile_read(int which, double* total_ms, long* launches, double* flop, int* segments);

/* Deferred weight gradients (no reference counterpart; off by default).  While on, pg_decoder_bwd enqueues the decoder's
 * large weight-gradient GEMMs, alternating, on TWO library-owned side streams (created on first use, never destroyed; the
 * second also takes the small embedding / z-projection / decoder_lat gradients) so that they run beside the dz chain and
 * the following pg_encoder_bwd, a latency-bound kernel on a quarter of the CUs.  The decoder weight gradients are
 * complete, in the order of the caller's stream, after the next pg_encoder_bwd or pg_clip_adam on that stream, or after
 * pg_deferred_join(stream) (e.g. the stream a gradient all-reduce will run on).  A caller that reads the gradients
 * without any of the three must leave this off.  While on, the forward calls use the first side stream too:
 * pg_prepare_weights rebuilds the decoder's shadow weights there, and when pg_encoder_fwd and pg_decoder_fwd follow it with
 * the SAME tokens pointer and workspace, the encoder's token sort and the decoder's packing + token sort run there beside the
 * encoder recurrence (ordered behind an event that pg_encoder_fwd records on the caller's stream once the tokens are packed,
 * joined into the caller's stream by pg_decoder_fwd / pg_encoder_bwd); any other call order is served on the caller's stream.
 * pg_defer_weight_grads returns the previous setting (on < 0: query). */
PG_API int pg_defer_weight_grads(int on);
PG_API int pg_deferred_join(void* stream);

/* Failure of a persistent recurrence kernel (no reference counterpart).  The group-persistent kernels' workgroups wait
 * for each other; every wait is bounded (pg_persist_spin_limit polls, default 2^21 = 2-4 s), and a workgroup that gives up
 * sets a sticky device word.  Device side: pg_clip_adam's kernel reads the word and SKIPS the update, so garbage
 * gradients never reach the weights.  Host side: the word is copied to its pinned mirror behind every persistent launch -
 * in a training step (pg_decoder_fwd with save_for_backward, then pg_decoder_bwd) ONCE, at the end of pg_decoder_bwd - and
 * every entry point checks the mirror on entry (an early warning); pg_check_failure(stream) synchronises `stream` and reads
 * the device word itself (call it before trusting or saving results).  The call that
 * detects a failure returns PG_ERR_PERSIST once; the library then clears the word and switches every recurrence to the
 * per-step kernels (as pg_persist_disable(15)), so later calls work.  pg_persist_spin_limit(limit) sets the poll bound and
 * returns the previous one (limit < 0: query; 0 makes every unsatisfied wait give up at once - the test hook). */
PG_API int pg_check_failure(void* stream);
PG_API unsigned pg_persist_spin_limit(long limit);
/* Data parallelism (no reference counterpart): the give-up word is per process, while a failing rank's gradients have
 * already been summed into every rank.  pg_fail_flag_export copies the device word to dst[0] (device memory) behind the
 * work queued on `stream`; the caller MAX-all-reduces that int over the ranks and pg_fail_flag_import writes the result
 * back (and refreshes the pinned mirror) BEFORE pg_clip_adam, so that EVERY rank skips the update and EVERY rank's next
 * pg_check_failure returns PG_ERR_PERSIST. */
PG_API int pg_fail_flag_export(int32_t* dst, void* stream);
PG_API int pg_fail_flag_import(const int32_t* src, void* stream);
/* pg_entry_checks(0): the entry points stop polling the mirror on entry (a rank that raised in the middle of a step
 * would leave its peers in the step's all-reduce); pg_check_failure still reports.  The data-parallel trainer switches
 * them off for an epoch and checks at its end, on every rank.  Returns the previous setting (on < 0: query). */
PG_API int pg_entry_checks(int on);

/* Test / A-B hook (no reference counterpart): the recurrences run as persistent kernels where the shape is covered
 * (bf16, Hq = 256, Hd = 512); mask bits force the per-step kernels instead: 1 encoder, 2 decoder forward, 4 decoder BPTT,
 * 8 the sampler (pg_sample then launches 3 cells + 1 head kernel per token instead of one kernel per batch).
 * mask < 0 only queries.  Returns the previous mask.  The environment variable PG_NO_PERSIST sets the initial mask. */
PG_API int pg_persist_disable(int mask);

#ifdef __cplusplus
}
#endif
#endif
