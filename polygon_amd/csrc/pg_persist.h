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

// true when the persistent kernels cover this hidden size (others take the per-step path)
bool enc_persist_supported(int H);
int launch_enc_fwd_persist(const EncFwdPersist& a, const Sched& sd, int H, hipStream_t stream);
int launch_enc_bwd_persist(const EncBwdPersist& a, const Sched& sd, int H, hipStream_t stream);
