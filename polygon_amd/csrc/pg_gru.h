// Recurrent GRU step kernels (forward cell with fused gate epilogue, backward cell with fused gate-gradient
// epilogue).  One launch = one time step of one layer over the active (length-sorted prefix) rows.
#pragma once
#include "pg_common.h"

enum { PG_GI_ARRAY = 0, PG_GI_TABLE = 1, PG_GI_FUSED = 2, PG_GI_DYN = 3 /* kernel-internal: read a.gi_mode at run time */ };

template <typename T> struct GruFwdArgs {
  int rows = 0, H = 0;
  // recurrent operand h_{t-1}: operand-dtype copy feeds the MFMA, fp32 copy feeds z*h_prev.  Both null => h_prev = 0.
  const T* hprev_op = nullptr;
  const float* hprev32 = nullptr;
  const T* Whh = nullptr;     // [3H, H] torch layout (r|z|n rows)
  const float* bhn = nullptr; // [H]  (b_hn; b_hr, b_hz are folded into the gi source)
  // gi = W_ih x + b_ih (+ b_hr | b_hz | 0) source
  int gi_mode = PG_GI_ARRAY;
  const float* gi = nullptr;   // ARRAY: [rows, 3H]
  const float* table = nullptr; // TABLE: [V, 3H] rows gathered by tok[]
  const int* tok = nullptr;     // TABLE: [rows]
  const float* zproj = nullptr; // TABLE: optional [rows, 3H] added per row (decoder layer 0: W_ih[:, E:] z)
  const T* x = nullptr;         // FUSED: [rows, Kx]
  int Kx = 0;
  const T* Wih = nullptr;       // FUSED: [3H, Kx]
  const float* bias_gi = nullptr; // FUSED: [3H]
  // outputs (all [rows, H], ld = H)
  T* h_op = nullptr;
  float* h32 = nullptr;
  T* sv_r = nullptr; T* sv_z = nullptr; T* sv_n = nullptr; T* sv_hn = nullptr; T* sv_hprev = nullptr;  // training saves
  T* h_drop = nullptr;          // h * inter-layer dropout mask (next layer's input), nullable
  float p_drop = 0.0f;
  unsigned long long seed = 0;
  int layer = 0, t = 0, b_mul = 1, b_add = 0;
  int dbg = 0;  // timing experiments only: 1 = skip main loop, 2 = skip epilogue, 4 = skip stores
};

template <typename T> struct GruBwdArgs {
  int rows_out = 0, rows_in = 0, H = 0;
  const T* dGH_in = nullptr;     // [rows_in, 3H]: gate gradients (hidden side) of step t+1
  const T* WhhT = nullptr;       // [H, 3H] = W_hh^T
  const float* dhz_in = nullptr; // [rows_in, H]: dh_{t+1} * z_{t+1}
  const float* dout = nullptr;   // [rows_out, H]: gradient arriving from above at step t (nullable)
  const float* inject = nullptr; // [>=rows_out, H]: added for rows >= rows_in (encoder: gradient of the final state)
  const T* sv_r = nullptr; const T* sv_z = nullptr; const T* sv_n = nullptr; const T* sv_hn = nullptr;
  const T* sv_hprev = nullptr;   // saves of step t, [rows_out, H]
  T* dGI = nullptr; T* dGH = nullptr;  // [rows_out, 3H]
  float* dhz_out = nullptr;      // [rows_out, H]
  int final_step = 0;            // 1: only dh0 (+)= acc + dhz_in for rows_out rows
  float* dh0 = nullptr;
  int dh0_accumulate = 0;
  // wavefront mode: the gradient from the layer above is computed here instead of arriving in `dout`:
  //   dout_t = dropmask(layer, t) * (dGI_up[rows_out, 3H] . WihT_up[H, 3H]^T)
  const T* dGI_up = nullptr;
  const T* WihT_up = nullptr;
  float p_drop = 0.0f;
  unsigned long long seed = 0;
  int layer = 0, t = 0, b_mul = 1, b_add = 0;
  int dbg = 0;
};

// One launch = one anti-diagonal of the (layer, t) grid of the decoder: up to PG_MAX_LAYERS independent cells.
template <typename T> struct GruFwdDiag {
  int n = 0;
  GruFwdArgs<T> a[8];
};
template <typename T> struct GruBwdDiag {
  int n = 0;
  GruBwdArgs<T> a[8];
};
template <typename T> int launch_gru_fwd_diag(const GruFwdDiag<T>& d, hipStream_t stream);
template <typename T> int launch_gru_bwd_diag(const GruBwdDiag<T>& d, hipStream_t stream);

template <typename T> int launch_gru_fwd(const GruFwdArgs<T>& a, hipStream_t stream);
template <typename T> int launch_gru_bwd(const GruBwdArgs<T>& a, hipStream_t stream);
