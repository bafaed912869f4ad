// VAE.forward_encoder for every encoder shape the reference can build (vae_model.py:63-81: q_n_layers, q_bidir,
// q_dropout) and its BPTT, on the per-step GRU kernels (pg_gru.hip) and the batched GEMM kernels (pg_gemm.hip).
//
// Packed, time-major batch as everywhere else (rows of step t = the prefix b < B_t at packed rows off[t]..).  A layer's
// two directions are independent until the next layer reads both, so one launch carries a step of each:
//   forward direction, step t: rows [0, B_t), h_{t-1} = the layer's own output at off[t-1] (the prefix of a longer step);
//   reverse direction, step t (walking t = T-1 .. 0): rows [0, B_{t+1}) continue from their state at step t+1, rows
//     [B_{t+1}, B_t) are sequences whose LAST token is t and start from zero - what nn.GRU(bidirectional=True) does on a
//     PackedSequence (each sequence is reversed within its own length).  Two cells of the same launch.
// Both write their output at packed row off[t] + b, so layer l+1 reads [forward | reverse] (concat + nn.GRU's
// inter-layer dropout, one elementwise kernel) and the heads read the two final states (vae_model.py:228-229).
// BPTT mirrors it: the forward direction walks t = T-1 .. 0 as the default encoder does; the reverse direction walks
// t = 0 .. T-1, where the rows of step t are a prefix of the rows of step t-1 (rows_in = rows_out = B_t).
//
// The default shape (one layer, one direction) has its own path in pg_vae.hip (gather table + persistent kernels); this
// file is the general one and keeps operand copies of the encoder weights in its own workspace, rebuilt per forward call.
#include <math.h>
#include <type_traits>
#include "pg_gemm.h"
#include "pg_gru.h"
#include "pg_host.h"
#include "pg_layout.h"
#include "pg_misc.h"

namespace {

struct EncGenWs {
  size_t tok, tgt, rt, rb, scalars, dT, slab, slab_bytes;
  size_t whh[PG_MAX_ENC_LAYERS][2], whhT[PG_MAX_ENC_LAYERS][2], wih[PG_MAX_ENC_LAYERS][2], wihT[PG_MAX_ENC_LAYERS][2];
  size_t bias[PG_MAX_ENC_LAYERS][2], table[2];
  size_t out[PG_MAX_ENC_LAYERS][2], sv[PG_MAX_ENC_LAYERS][2][5], carry[PG_MAX_ENC_LAYERS][2];
  size_t X[PG_MAX_ENC_LAYERS];   // input of layer l >= 1: [N, Hq*dirs], dropped
  size_t gi[2];
  size_t hfin_op, wq, wqT, bq, mulv, eps;
  size_t dq, dq_op, dhfin[2], dGI[2], dGH[2], dhz[2][2], dX, dout[2];
  size_t total;
};

EncGenWs encgen_layout(const pg_dims* d, int dtype, long B, long N) {
  const size_t es = dtype == PG_BF16 ? 2 : 4;
  EncGenWs w;
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t r = p; p = align_up(p + bytes, 256); return r; };
  const size_t H = d->Hq, Dz = d->Dz, Vp = vpad(d), V = d->V;
  const int Lq = enc_layers(d), nd = enc_dirs(d);
  const size_t Hd = H * nd;
  w.tok = take(N * 4); w.tgt = take(N * 4); w.rt = take(N * 4); w.rb = take(N * 4);
  w.scalars = take(SC_N * 4);
  w.dT = take(Vp * 3 * H * 4);
  w.slab_bytes = tok_scatter_layout(N, 3 * (int)H, (int)Vp).total; w.slab = take(w.slab_bytes);
  for (int l = 0; l < Lq; ++l) {
    const size_t in = enc_in(d, l);
    for (int r = 0; r < nd; ++r) {
      w.whh[l][r] = take(3 * H * H * es); w.whhT[l][r] = take(3 * H * H * es);
      w.wih[l][r] = l ? take(3 * H * in * es) : 0; w.wihT[l][r] = l ? take(3 * H * in * es) : 0;
      w.bias[l][r] = l ? take(3 * H * 4) : 0;
      w.out[l][r] = take(N * H * es);
      for (int i = 0; i < 5; ++i) w.sv[l][r][i] = take(N * H * es);
      w.carry[l][r] = take(B * H * 4);
    }
    w.X[l] = l ? take(N * in * es) : 0;
  }
  for (int r = 0; r < nd; ++r) {
    w.table[r] = take(V * 3 * H * 4);
    w.gi[r] = Lq > 1 ? take(N * 3 * H * 4) : 0;
    w.dhfin[r] = take(B * H * 4);
    w.dGI[r] = take(N * 3 * H * es); w.dGH[r] = take(N * 3 * H * es);
    w.dhz[r][0] = take(B * H * 4); w.dhz[r][1] = take(B * H * 4);
    w.dout[r] = Lq > 1 ? take(N * H * 4) : 0;
  }
  w.hfin_op = take(B * Hd * es); w.wq = take(2 * Dz * Hd * es); w.wqT = take(2 * Dz * Hd * es); w.bq = take(2 * Dz * 4);
  w.mulv = take(B * 2 * Dz * 4); w.eps = take(B * Dz * 4);
  w.dq = take(B * 2 * Dz * 4); w.dq_op = take(B * 2 * Dz * es);
  w.dX = Lq > 1 ? take(N * Hd * 4) : 0;
  w.total = p;
  return w;
}

// X[r, :] = mask(layer, t(r), global row(r), :) * [out_f[r, :] | out_b[r, :]]      (nn.GRU inter-layer dropout on the
// concatenated output of a layer; mask = 1 when p == 0).  One thread = 4 columns (the granule of the mask hash).
template <typename T>
__global__ void enc_concat_drop_kernel(const T* __restrict__ of, const T* __restrict__ ob, long N, int H, int nd,
                                       const int* __restrict__ rt, const int* __restrict__ rb, float p, unsigned long long seed,
                                       int layer, int b_mul, int b_add, T* __restrict__ X) {
  const int W = H * nd, W4 = W / 4;
  const float inv_keep = p > 0.0f ? 1.0f / (1.0f - p) : 1.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N * W4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / W4;
    const int c4 = (int)(i % W4), c = c4 * 4;
    const T* src = c < H ? of + r * H + c : ob + r * H + (c - H);
    float m[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (p > 0.0f) dropout_scale4(seed, layer, rt[r], rb[r] * b_mul + b_add, c4, p, inv_keep, m);
#pragma unroll
    for (int e = 0; e < 4; ++e) X[r * W + c + e] = from_f32<T>(to_f32<T>(src[e]) * m[e]);
  }
}

// gradient of the kernel above: dout_f / dout_b [N, H] fp32 = mask * dX[:, 0:H] / dX[:, H:2H]
__global__ void enc_split_drop_kernel(const float* __restrict__ dX, long N, int H, int nd, const int* __restrict__ rt,
                                      const int* __restrict__ rb, float p, unsigned long long seed, int layer, int b_mul,
                                      int b_add, float* __restrict__ df, float* __restrict__ db) {
  const int W = H * nd, W4 = W / 4;
  const float inv_keep = p > 0.0f ? 1.0f / (1.0f - p) : 1.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N * W4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / W4;
    const int c4 = (int)(i % W4), c = c4 * 4;
    float* dst = c < H ? df + r * H + c : db + r * H + (c - H);
    float m[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (p > 0.0f) dropout_scale4(seed, layer, rt[r], rb[r] * b_mul + b_add, c4, p, inv_keep, m);
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[e] = dX[r * W + c + e] * m[e];
  }
}

// h [B, H*nd] (operand precision) = [carry_f | carry_b]: the last layer's final states, forward then reverse
template <typename T>
__global__ void enc_concat_final_kernel(const float* __restrict__ cf, const float* __restrict__ cb, int B, int H, int nd,
                                        T* __restrict__ out) {
  const int W = H * nd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)B * W; i += (long)gridDim.x * blockDim.x) {
    const long b = i / W;
    const int c = (int)(i % W);
    out[i] = from_f32<T>(c < H ? cf[b * H + c] : cb[b * H + (c - H)]);
  }
}

struct Ctx {
  ParamOff po;
  EncGenWs w;
  Sched se;
  long N;
  int Lq, nd, H, Hd;
};

int make_ctx(const pg_dims* d, int dtype, size_t ws_bytes, int B, int Tmax, const int32_t* lens, const char* who, Ctx* c) {
  c->po = param_offsets(d);
  PG_TRY(make_sched(lens, B, Tmax, 0, &c->se));
  c->N = c->se.off[c->se.T];
  c->w = encgen_layout(d, dtype, B, c->N);
  PG_REQUIRE(ws_bytes >= c->w.total, "%s: workspace too small (%zu < %zu)", who, ws_bytes, c->w.total);
  c->Lq = enc_layers(d); c->nd = enc_dirs(d); c->H = d->Hq; c->Hd = d->Hq * c->nd;
  return PG_OK;
}

template <typename T>
int encgen_fwd_t(const pg_dims* d, int dtype, const float* params, void* ws, size_t ws_bytes, int B, int Tmax,
                 const int32_t* d_tokens, const int32_t* lens, const float* eps, const pg_dp* dp, float p_drop, uint64_t seed,
                 int save, float* z, float* mu, float* logvar, float* kl, hipStream_t st) {
  Ctx c;
  PG_TRY(make_ctx(d, dtype, ws_bytes, B, Tmax, lens, "pg_encoder_gen_fwd", &c));
  const ParamOff& po = c.po;
  const EncGenWs& w = c.w;
  const Sched& se = c.se;
  const int Lq = c.Lq, nd = c.nd, H = c.H, Hd = c.Hd, Dz = d->Dz, E = d->E, V = d->V;
  const long N = c.N;
  const int b_mul = dp ? dp->b_mul : 1, b_add = dp ? dp->b_add : 0;
  if (Lq == 1) p_drop = 0.0f;   // nn.GRU applies dropout between layers only (vae_model.py:69)

  // ---- operand copies of the encoder's weights (this call's parameters)
  {
    CopyTable tab;
    tab.n = 0;
    long maxel = 0;
    auto add = [&](long src, size_t dst, int rows, int cols, int src_ld, int dst_ld, int tr, int f32) {
      CopyDesc cd = {src, dst, rows, cols, src_ld, dst_ld, tr, f32};
      tab.d[tab.n++] = cd;
      maxel = std::max(maxel, (long)rows * cols);
    };
    for (int l = 0; l < Lq; ++l)
      for (int r = 0; r < nd; ++r) {
        const int in = enc_in(d, l);
        add(po.g_whh[l][r], w.whh[l][r], 3 * H, H, H, H, 0, 0);
        add(po.g_whh[l][r], w.whhT[l][r], 3 * H, H, H, 3 * H, 1, 0);
        if (l) {
          add(po.g_wih[l][r], w.wih[l][r], 3 * H, in, in, in, 0, 0);
          add(po.g_wih[l][r], w.wihT[l][r], 3 * H, in, in, 3 * H, 1, 0);
        }
      }
    add(po.qmu_w, w.wq, Dz, Hd, Hd, Hd, 0, 0);
    add(po.qlv_w, w.wq + (size_t)Dz * Hd * sizeof(T), Dz, Hd, Hd, Hd, 0, 0);
    add(po.qmu_w, w.wqT, Dz, Hd, Hd, 2 * Dz, 1, 0);
    add(po.qlv_w, w.wqT + (size_t)Dz * sizeof(T), Dz, Hd, Hd, 2 * Dz, 1, 0);
    add(po.qmu_b, w.bq, 1, Dz, Dz, Dz, 0, 1);
    add(po.qlv_b, w.bq + (size_t)Dz * 4, 1, Dz, Dz, Dz, 0, 1);
    PG_REQUIRE(tab.n <= PG_MAX_COPY, "copy table overflow");
    hipLaunchKernelGGL(copy_table_kernel<T>, dim3(std::min<int>(cdiv(maxel, 256), 256), tab.n), dim3(256), 0, st, params,
                       static_cast<char*>(ws), tab);
    PG_CHECK_LAUNCH();
    for (int r = 0; r < nd; ++r) {
      PG_TRY(launch_gi_table(params + po.x_emb, E, params + po.g_wih[0][r], E, params + po.g_bih[0][r], params + po.g_bhh[0][r], H, V,
                             at<float>(ws, w.table[r]), st));
      for (int l = 1; l < Lq; ++l) {
        hipLaunchKernelGGL(bias_fold_kernel, dim3(cdiv(3 * H, 256)), dim3(256), 0, st, params + po.g_bih[l][r],
                           params + po.g_bhh[l][r], H, at<float>(ws, w.bias[l][r]));
        PG_CHECK_LAUNCH();
      }
    }
  }
  hipLaunchKernelGGL(build_packed_kernel<T>, dim3(se.T, cdiv(B, 256)), dim3(256), 0, st, d_tokens, B, Tmax, d->pad, se, 1,
                     at<int>(ws, w.tok), at<int>(ws, w.tgt), at<int>(ws, w.rt), at<int>(ws, w.rb), (int*)nullptr,
                     (unsigned*)nullptr, 0);
  PG_CHECK_LAUNCH();
  float* scal = at<float>(ws, w.scalars);
  PG_CHECK_HIP(hipMemsetAsync(scal + SC_KL_SUM, 0, 2 * sizeof(float), st));

  for (int l = 0; l < Lq; ++l) {
    const int in = enc_in(d, l);
    const T* X = l ? at<T>(ws, w.X[l]) : nullptr;
    if (l)
      for (int r = 0; r < nd; ++r) {   // gi = X . W_ih^T + b_ih + (b_hr | b_hz | 0) for every packed row at once
        GemmEpi<T> e;
        e.bias = at<float>(ws, w.bias[l][r]); e.C32 = at<float>(ws, w.gi[r]); e.ldc = 3 * H;
        PG_TRY(launch_gemm_nt<T>(X, in, at<T>(ws, w.wih[l][r]), in, (int)N, 3 * H, in, e, st));
      }
    for (int i = 0; i < se.T; ++i) {
      GruFwdDiag<T> dg;
      dg.n = 0;
      auto cell = [&](int r, int t, int r0, int rows, bool has_prev, int t_prev) {
        if (rows <= 0) return;
        GruFwdArgs<T>& a = dg.a[dg.n++];
        a = GruFwdArgs<T>();
        a.rows = rows; a.H = H;
        float* carry = at<float>(ws, w.carry[l][r]) + (long)r0 * H;
        T* out = at<T>(ws, w.out[l][r]);
        if (has_prev) { a.hprev_op = out + ((long)se.off[t_prev] + r0) * H; a.hprev32 = carry; }
        a.Whh = at<T>(ws, w.whh[l][r]);
        a.bhn = params + po.g_bhh[l][r] + 2 * H;
        const long row = (long)se.off[t] + r0;
        if (l == 0) { a.gi_mode = PG_GI_TABLE; a.table = at<float>(ws, w.table[r]); a.tok = at<int>(ws, w.tok) + row; }
        else { a.gi_mode = PG_GI_ARRAY; a.gi = at<float>(ws, w.gi[r]) + row * 3 * H; }
        a.h_op = out + row * H;
        a.h32 = carry;
        if (save) {
          a.sv_r = at<T>(ws, w.sv[l][r][0]) + row * H; a.sv_z = at<T>(ws, w.sv[l][r][1]) + row * H;
          a.sv_n = at<T>(ws, w.sv[l][r][2]) + row * H; a.sv_hn = at<T>(ws, w.sv[l][r][3]) + row * H;
          a.sv_hprev = at<T>(ws, w.sv[l][r][4]) + row * H;
        }
      };
      cell(0, i, 0, sched_B(se, i), i > 0, i - 1);
      if (nd == 2) {
        const int t = se.T - 1 - i;
        const int Bn = t + 1 < se.T ? sched_B(se, t + 1) : 0;   // rows that already ran step t+1
        cell(1, t, 0, Bn, true, t + 1);
        cell(1, t, Bn, sched_B(se, t) - Bn, false, 0);
      }
      PG_TRY(launch_gru_fwd_diag<T>(dg, st));
    }
    if (l + 1 < Lq) {
      if (nd == 1 && p_drop <= 0.0f) {
        PG_CHECK_HIP(hipMemcpyAsync(at<T>(ws, w.X[l + 1]), at<T>(ws, w.out[l][0]), (size_t)N * H * sizeof(T), hipMemcpyDeviceToDevice, st));
      } else {
        hipLaunchKernelGGL(enc_concat_drop_kernel<T>, dim3(ew_blocks(N * Hd / 4)), dim3(256), 0, st, at<T>(ws, w.out[l][0]),
                           at<T>(ws, w.out[l][nd - 1]), N, H, nd, at<int>(ws, w.rt), at<int>(ws, w.rb), p_drop,
                           (unsigned long long)seed, l, b_mul, b_add, at<T>(ws, w.X[l + 1]));
        PG_CHECK_LAUNCH();
      }
    }
  }
  // ---- heads: q_mu | q_logvar on [h_forward | h_reverse] of the last layer (vae_model.py:228-231), z and KL
  hipLaunchKernelGGL(enc_concat_final_kernel<T>, dim3(ew_blocks((long)B * Hd)), dim3(256), 0, st, at<float>(ws, w.carry[Lq - 1][0]),
                     at<float>(ws, w.carry[Lq - 1][nd - 1]), B, H, nd, at<T>(ws, w.hfin_op));
  PG_CHECK_LAUNCH();
  const float inv_B = (dp && dp->kl_inv_batch > 0.0f) ? dp->kl_inv_batch : 1.0f / (float)B;
  if constexpr (std::is_same<T, bf16_t>::value) {   // heads + reparameterisation + KL as one kernel (pg_heads.hip)
    int status = PG_OK;
    if (launch_enc_heads(at<T>(ws, w.hfin_op), B, Hd, at<T>(ws, w.wq), at<float>(ws, w.bq), Dz, eps, at<float>(ws, w.mulv), z, (T*)nullptr,
                         mu, logvar, save ? at<float>(ws, w.eps) : (float*)nullptr, scal + SC_KL_SUM, inv_B, kl, st, &status))
      return status;
  }
  GemmEpi<T> e;
  e.bias = at<float>(ws, w.bq); e.C32 = at<float>(ws, w.mulv); e.ldc = 2 * Dz;
  PG_TRY(launch_gemm_nt<T>(at<T>(ws, w.hfin_op), Hd, at<T>(ws, w.wq), Hd, B, 2 * Dz, Hd, e, st));
  hipLaunchKernelGGL(reparam_kl_kernel<T>, dim3(std::min(cdiv((long)B * Dz, 256), 64)), dim3(256), 0, st, at<float>(ws, w.mulv), eps, B, Dz, z,
                     (T*)nullptr, mu, logvar, save ? at<float>(ws, w.eps) : (float*)nullptr, scal + SC_KL_SUM, inv_B, kl);
  PG_CHECK_LAUNCH();
  return PG_OK;
}

template <typename T>
int encgen_bwd_t(const pg_dims* d, int dtype, const float* params, void* ws, size_t ws_bytes, int B, int Tmax,
                 const int32_t* lens, const pg_dp* dp, float p_drop, uint64_t seed, const float* dz, float g_host,
                 const float* g_dev, float* grads, hipStream_t st) {
  Ctx c;
  PG_TRY(make_ctx(d, dtype, ws_bytes, B, Tmax, lens, "pg_encoder_gen_bwd", &c));
  const ParamOff& po = c.po;
  const EncGenWs& w = c.w;
  const Sched& se = c.se;
  const int Lq = c.Lq, nd = c.nd, H = c.H, Hd = c.Hd, Dz = d->Dz, E = d->E, V = d->V, Vp = vpad(d);
  const long N = c.N;
  const int b_mul = dp ? dp->b_mul : 1, b_add = dp ? dp->b_add : 0;
  if (Lq == 1) p_drop = 0.0f;
  float* scal = at<float>(ws, w.scalars);
  const float inv_B = (dp && dp->kl_inv_batch > 0.0f) ? dp->kl_inv_batch : 1.0f / (float)B;
  float* dq = at<float>(ws, w.dq);
  T* dq_op = at<T>(ws, w.dq_op);
  hipLaunchKernelGGL(reparam_bwd_kernel<T>, dim3(cdiv((long)B * Dz, 256)), dim3(256), 0, st, at<float>(ws, w.mulv),
                     at<float>(ws, w.eps), dz, g_host * inv_B, g_dev, B, Dz, dq, dq_op);
  PG_CHECK_LAUNCH();
  const T* hfin = at<T>(ws, w.hfin_op);
  PG_TRY(launch_gemm_tn<T>(dq_op, 2 * Dz, hfin, Hd, Dz, Hd, B, grads + po.qmu_w, Hd, 1.0f, nullptr, st, grads + po.qmu_b));
  PG_TRY(launch_gemm_tn<T>(dq_op + Dz, 2 * Dz, hfin, Hd, Dz, Hd, B, grads + po.qlv_w, Hd, 1.0f, nullptr, st, grads + po.qlv_b));
  for (int r = 0; r < nd; ++r) {   // gradient of each direction's final state: its H rows of [W_mu ; W_logvar]^T
    GemmEpi<T> e;
    e.C32 = at<float>(ws, w.dhfin[r]); e.ldc = H;
    PG_TRY(launch_gemm_nt<T>(dq_op, 2 * Dz, at<T>(ws, w.wqT) + (long)r * H * 2 * Dz, 2 * Dz, B, H, 2 * Dz, e, st));
  }

  for (int l = Lq - 1; l >= 0; --l) {
    const bool top = l == Lq - 1;
    const int in = enc_in(d, l);
    for (int i = 0; i < se.T; ++i) {
      GruBwdDiag<T> dg;
      dg.n = 0;
      auto cell = [&](int r, int t, int rows_in, int t_next, bool inject) {
        GruBwdArgs<T>& a = dg.a[dg.n++];
        a = GruBwdArgs<T>();
        a.H = H;
        a.rows_out = sched_B(se, t);
        a.rows_in = rows_in;
        a.dGH_in = at<T>(ws, w.dGH[r]) + (long)se.off[rows_in > 0 ? t_next : 0] * 3 * H;
        a.WhhT = at<T>(ws, w.whhT[l][r]);
        a.dhz_in = at<float>(ws, w.dhz[r][t_next & 1]);
        a.inject = (top && inject) ? at<float>(ws, w.dhfin[r]) : nullptr;
        a.dout = top ? nullptr : at<float>(ws, w.dout[r]) + (long)se.off[t] * H;
        const long o = (long)se.off[t] * H;
        a.sv_r = at<T>(ws, w.sv[l][r][0]) + o; a.sv_z = at<T>(ws, w.sv[l][r][1]) + o; a.sv_n = at<T>(ws, w.sv[l][r][2]) + o;
        a.sv_hn = at<T>(ws, w.sv[l][r][3]) + o; a.sv_hprev = at<T>(ws, w.sv[l][r][4]) + o;
        a.dGI = at<T>(ws, w.dGI[r]) + o * 3; a.dGH = at<T>(ws, w.dGH[r]) + o * 3;
        a.dhz_out = at<float>(ws, w.dhz[r][t & 1]);
      };
      {   // forward direction: t = T-1 .. 0; rows that stop at t take the final-state gradient
        const int t = se.T - 1 - i;
        cell(0, t, t + 1 < se.T ? sched_B(se, t + 1) : 0, t + 1, true);
      }
      if (nd == 2)   // reverse direction: t = 0 .. T-1; its final state is the one after step 0, every row has one
        cell(1, i, i > 0 ? sched_B(se, i) : 0, i > 0 ? i - 1 : 1, i == 0);
      PG_TRY(launch_gru_bwd_diag<T>(dg, st));
    }
    for (int r = 0; r < nd; ++r) {
      const T* dGI = at<T>(ws, w.dGI[r]);
      const T* dGH = at<T>(ws, w.dGH[r]);
      PG_TRY(launch_gemm_tn<T>(dGH, 3 * H, at<T>(ws, w.sv[l][r][4]), H, 3 * H, H, (int)N, grads + po.g_whh[l][r], H, 1.0f, nullptr, st,
                               grads + po.g_bhh[l][r]));
      if (l) {
        PG_TRY(launch_gemm_tn<T>(dGI, 3 * H, at<T>(ws, w.X[l]), in, 3 * H, in, (int)N, grads + po.g_wih[l][r], in, 1.0f, nullptr, st,
                                 grads + po.g_bih[l][r]));
        GemmEpi<T> e;   // dX (+)= dGI . W_ih
        e.C32 = at<float>(ws, w.dX); e.ldc = in; e.accumulate = r > 0;
        PG_TRY(launch_gemm_nt<T>(dGI, 3 * H, at<T>(ws, w.wihT[l][r]), 3 * H, (int)N, in, 3 * H, e, st));
      } else {
        // embedding layer: dTable = scatter-sum of dGI by token, then x_emb and W_ih gradients from the V-row table (as pg_encoder_bwd)
        float* dT = at<float>(ws, w.dT);
        PG_TRY(launch_tok_scatter_sum<T>(at<int>(ws, w.tok), dGI, 3 * H, N, 3 * H, Vp, dT, grads + po.g_bih[0][r],
                                         at<char>(ws, w.slab), w.slab_bytes, st));
        PG_TRY(small_gemm(dT, 3 * H, 1, params + po.g_wih[0][r], E, 1, V, E, 3 * H, grads + po.x_emb, E, d->pad, st));
        PG_TRY(small_gemm(dT, 1, 3 * H, params + po.x_emb, E, 1, 3 * H, E, V, grads + po.g_wih[0][r], E, -1, st));
      }
    }
    if (l) {
      hipLaunchKernelGGL(enc_split_drop_kernel, dim3(ew_blocks(N * Hd / 4)), dim3(256), 0, st, at<float>(ws, w.dX), N, H, nd,
                         at<int>(ws, w.rt), at<int>(ws, w.rb), p_drop, (unsigned long long)seed, l - 1, b_mul, b_add,
                         at<float>(ws, w.dout[0]), at<float>(ws, w.dout[nd - 1]));
      PG_CHECK_LAUNCH();
    }
  }
  // as pg_encoder_bwd: the decoder's deferred weight gradients are complete in the order of this stream from here on
  // (a following forward pass on the same workspace may overwrite what they read)
  PG_TRY(pg_deferred_join(st));
  return PG_OK;
}

int check_gen(const pg_dims* d, int dtype, float p_drop, const char* who) {
  PG_TRY(check_dims(d, dtype));
  PG_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "%s: dropout %f outside [0,1)", who, p_drop);
  PG_REQUIRE(d->Hq % 32 == 0, "%s: Hq must be a multiple of 32", who);
  return PG_OK;
}

}  // namespace

extern "C" size_t pg_encoder_gen_workspace_bytes(const pg_dims* d, int dtype, int B, long n_tokens) {
  if (check_dims(d, dtype) != PG_OK || B < 1 || n_tokens < B) return 0;
  return encgen_layout(d, dtype, B, n_tokens).total;
}

extern "C" int pg_encoder_gen_fwd(const pg_dims* d, int dtype, const float* params, void* ws, size_t ws_bytes, int B, int Tmax,
                                  const int32_t* d_tokens, const int32_t* h_lengths, const float* eps, const pg_dp* dp,
                                  float p_drop, uint64_t seed, int save_for_backward, float* z, float* mu, float* logvar,
                                  float* kl, void* stream) {
  PG_TRY(check_gen(d, dtype, p_drop, "pg_encoder_gen_fwd"));
  PG_TRY(check_persist());
  PG_REQUIRE(params && ws && d_tokens && z, "pg_encoder_gen_fwd: NULL buffer");
  return dtype == PG_BF16 ? encgen_fwd_t<bf16_t>(d, dtype, params, ws, ws_bytes, B, Tmax, d_tokens, h_lengths, eps, dp, p_drop, seed,
                                                 save_for_backward, z, mu, logvar, kl, (hipStream_t)stream)
                          : encgen_fwd_t<float>(d, dtype, params, ws, ws_bytes, B, Tmax, d_tokens, h_lengths, eps, dp, p_drop, seed,
                                                save_for_backward, z, mu, logvar, kl, (hipStream_t)stream);
}

extern "C" int pg_encoder_gen_bwd(const pg_dims* d, int dtype, const float* params, void* ws, size_t ws_bytes, int B, int Tmax,
                                  const int32_t* h_lengths, const pg_dp* dp, float p_drop, uint64_t seed, const float* dz,
                                  float g_host, const float* g_dev, float* grads, void* stream) {
  PG_TRY(check_gen(d, dtype, p_drop, "pg_encoder_gen_bwd"));
  PG_TRY(check_persist());
  PG_REQUIRE(params && ws && grads, "pg_encoder_gen_bwd: NULL buffer");
  return dtype == PG_BF16 ? encgen_bwd_t<bf16_t>(d, dtype, params, ws, ws_bytes, B, Tmax, h_lengths, dp, p_drop, seed, dz, g_host,
                                                 g_dev, grads, (hipStream_t)stream)
                          : encgen_bwd_t<float>(d, dtype, params, ws, ws_bytes, B, Tmax, h_lengths, dp, p_drop, seed, dz, g_host,
                                                g_dev, grads, (hipStream_t)stream);
}
