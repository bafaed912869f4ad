// GRU recurrent step kernels for gfx950: gate GEMM on MFMA + fused gate (or gate-gradient) epilogue.
#include "pg_gru.h"
#include "pg_mma.h"

// ------------------------------------------------------------------------------------------------
// forward: h_t = GRUCell(gi_t, h_{t-1})
//   block tile = BM rows x BH hidden units x 3 gates; accumulators per wave: r, z, hn (+ in when the
//   input projection is fused).  torch.nn.GRU semantics:
//     r = s(gi_r + W_hr h + b_hr), z = s(gi_z + W_hz h + b_hz), n = tanh(gi_n + r*(W_hn h + b_hn)),
//     h' = (1-z) n + z h       (b_hr, b_hz live in the gi source, b_hn in bhn)
// ------------------------------------------------------------------------------------------------
template <class Cfg, int MODE>
__global__ __launch_bounds__(Cfg::NTHREADS) void gru_fwd_kernel(GruFwdArgs<typename Cfg::T> a) {
  using T = typename Cfg::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TU = Cfg::TU;
  constexpr int NACC = (MODE == PG_GI_FUSED ? 4 : 3) * TU;
  constexpr int BH = Cfg::BNG;
  const int H = a.H;
  const int j0 = blockIdx.x * BH, m0 = blockIdx.y * Cfg::BM;
  f32x16 acc[Cfg::TM][NACC];
{
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[i][j] = zacc;
  }
  if (a.hprev_op) {
    RowSrc<T> as{a.hprev_op, H, m0, a.rows};
    GateRowSrc<T> bs{a.Whh, H, H, j0, BH};
    mma_mainloop_nt<Cfg, NACC, 2>(acc, as, bs, H, smem);
  }
  if (MODE == PG_GI_FUSED) {
    RowSrc<T> as{a.x, a.Kx, m0, a.rows};
    GateRowSrc<T> bs{a.Wih, a.Kx, H, j0, BH};
    mma_mainloop_nt<Cfg, NACC, 3>(acc, as, bs, a.Kx, smem);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
#pragma unroll
  for (int tu = 0; tu < TU; ++tu) {
    const int j = j0 + Cfg::col_in_group(wn, tu, lane & 31);
    if (j >= H) continue;
    const float bhn = a.bhn[j];
    float bg_r = 0.f, bg_z = 0.f, bg_n = 0.f;
    if (MODE == PG_GI_FUSED) {
      bg_r = a.bias_gi[j]; bg_z = a.bias_gi[H + j]; bg_n = a.bias_gi[2 * H + j];
    }
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int b = m0 + Cfg::row_of(wm, tm, reg, lane);
        if (b >= a.rows) continue;
        float gr, gz, gn;
        if (MODE == PG_GI_ARRAY) {
          const float* g = a.gi + (long)b * 3 * H;
          gr = g[j]; gz = g[H + j]; gn = g[2 * H + j];
        } else if (MODE == PG_GI_TABLE) {
          const float* g = a.table + (long)a.tok[b] * 3 * H;
          gr = g[j]; gz = g[H + j]; gn = g[2 * H + j];
          if (a.zproj) {
            const float* zp = a.zproj + (long)b * 3 * H;
            gr += zp[j]; gz += zp[H + j]; gn += zp[2 * H + j];
          }
        } else {
          gr = bg_r; gz = bg_z; gn = bg_n + acc[tm][3 * TU + tu][reg];
        }
        const float r = sigmoidf_(gr + acc[tm][tu][reg]);
        const float z = sigmoidf_(gz + acc[tm][TU + tu][reg]);
        const float hn = acc[tm][2 * TU + tu][reg] + bhn;
        const float n = tanhf_(gn + r * hn);
        const long o = (long)b * H + j;
        const float hp = a.hprev32 ? a.hprev32[o] : 0.0f;
        const float h = (1.0f - z) * n + z * hp;
        a.h32[o] = h;
        a.h_op[o] = from_f32<T>(h);
        if (a.sv_r) {
          a.sv_r[o] = from_f32<T>(r);
          a.sv_z[o] = from_f32<T>(z);
          a.sv_n[o] = from_f32<T>(n);
          a.sv_hn[o] = from_f32<T>(hn);
          a.sv_hprev[o] = from_f32<T>(hp);
        }
        if (a.h_drop)
          a.h_drop[o] = from_f32<T>(h * dropout_scale(a.seed, a.layer, a.t, b * a.b_mul + a.b_add, j, a.p_drop, inv_keep));
      }
  }
}

// ------------------------------------------------------------------------------------------------
// backward: from dGH_{t+1} compute dh_t = dout_t + dh_{t+1} z_{t+1} + dGH_{t+1} W_hh, then the gate
// gradients of step t (oracle/vae_oracle.py:_gru_backward states the same equations on the CPU).
// ------------------------------------------------------------------------------------------------
template <class Cfg>
__global__ __launch_bounds__(Cfg::NTHREADS) void gru_bwd_kernel(GruBwdArgs<typename Cfg::T> a) {
  using T = typename Cfg::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H;
  const int j0 = blockIdx.x * Cfg::BN, m0 = blockIdx.y * Cfg::BM;
  f32x16 acc[Cfg::TM][Cfg::TN];
{
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = zacc;
  }
  if (a.rows_in > 0 && m0 < a.rows_in) {
    RowSrc<T> as{a.dGH_in, 3L * H, m0, a.rows_in};
    RowSrc<T> bs{a.WhhT, 3L * H, j0, H};
    mma_mainloop_nt<Cfg, Cfg::TN, 0>(acc, as, bs, 3 * H, smem);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
#pragma unroll
  for (int tn = 0; tn < Cfg::TN; ++tn) {
    const int j = j0 + Cfg::col_in_group(wn, tn, lane & 31);
    if (j >= H) continue;
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int b = m0 + Cfg::row_of(wm, tm, reg, lane);
        if (b >= a.rows_out) continue;
        const long o = (long)b * H + j;
        float dh = 0.0f;
        if (b < a.rows_in) dh = acc[tm][tn][reg] + a.dhz_in[o];
        else if (a.inject) dh = a.inject[o];
        if (a.final_step) {
          a.dh0[o] = a.dh0_accumulate ? a.dh0[o] + dh : dh;
          continue;
        }
        if (a.dout) dh += a.dout[o];
        const float r = to_f32<T>(a.sv_r[o]), z = to_f32<T>(a.sv_z[o]), n = to_f32<T>(a.sv_n[o]);
        const float hn = to_f32<T>(a.sv_hn[o]), hp = to_f32<T>(a.sv_hprev[o]);
        const float dnp = dh * (1.0f - z) * (1.0f - n * n);
        const float dzp = dh * (hp - n) * z * (1.0f - z);
        const float drp = dnp * hn * r * (1.0f - r);
        const long og = (long)b * 3 * H + j;
        a.dGI[og] = from_f32<T>(drp);
        a.dGI[og + H] = from_f32<T>(dzp);
        a.dGI[og + 2 * H] = from_f32<T>(dnp);
        a.dGH[og] = from_f32<T>(drp);
        a.dGH[og + H] = from_f32<T>(dzp);
        a.dGH[og + 2 * H] = from_f32<T>(dnp * r);
        a.dhz_out[o] = dh * z;
      }
  }
}

template <class Cfg, int MODE, typename T> static int launch_fwd_cfg(const GruFwdArgs<T>& a, hipStream_t stream) {
  dim3 grid(cdiv(a.H, Cfg::BNG), cdiv(a.rows, Cfg::BM));
  PG_LAUNCH_BIG_SMEM((gru_fwd_kernel<Cfg, MODE>), grid, dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, a);
  PG_CHECK_LAUNCH();
  return PG_OK;
}

template <typename T> int launch_gru_fwd(const GruFwdArgs<T>& a, hipStream_t stream) {
  constexpr int EPC = 16 / sizeof(T);
  if (a.rows <= 0) return PG_OK;
  PG_REQUIRE(a.H % EPC == 0 && a.H >= 32, "gru_fwd: hidden size %d must be a multiple of %d", a.H, EPC);
  PG_REQUIRE(a.gi_mode != PG_GI_FUSED || (a.Kx % EPC == 0), "gru_fwd: fused input width %d must be a multiple of %d", a.Kx, EPC);
  PG_REQUIRE((a.hprev_op == nullptr) == (a.hprev32 == nullptr), "gru_fwd: hprev_op / hprev32 must both be set or null");
  if (a.rows > 2048) {
    using Cfg = TileCfg<T, 128, 192, 4, 2, 2, 3>;
    if (a.gi_mode == PG_GI_ARRAY) return launch_fwd_cfg<Cfg, PG_GI_ARRAY, T>(a, stream);
    if (a.gi_mode == PG_GI_TABLE) return launch_fwd_cfg<Cfg, PG_GI_TABLE, T>(a, stream);
    return launch_fwd_cfg<Cfg, PG_GI_FUSED, T>(a, stream);
  }
  using Cfg = TileCfg<T, 64, 192, 8, 2, 2, 3>;
  if (a.gi_mode == PG_GI_ARRAY) return launch_fwd_cfg<Cfg, PG_GI_ARRAY, T>(a, stream);
  if (a.gi_mode == PG_GI_TABLE) return launch_fwd_cfg<Cfg, PG_GI_TABLE, T>(a, stream);
  return launch_fwd_cfg<Cfg, PG_GI_FUSED, T>(a, stream);
}

template <typename T> int launch_gru_bwd(const GruBwdArgs<T>& a, hipStream_t stream) {
  constexpr int EPC = 16 / sizeof(T);
  if (a.rows_out <= 0) return PG_OK;
  PG_REQUIRE(a.H % EPC == 0, "gru_bwd: hidden size %d must be a multiple of %d", a.H, EPC);
  PG_REQUIRE(a.rows_in <= a.rows_out, "gru_bwd: rows_in %d > rows_out %d (batch must be sorted by length)", a.rows_in, a.rows_out);
  using Cfg = TileCfg<T, 64, 64, 8, 2, 2, 1>;
  dim3 grid(cdiv(a.H, Cfg::BN), cdiv(a.rows_out, Cfg::BM));
  PG_LAUNCH_BIG_SMEM(gru_bwd_kernel<Cfg>, grid, dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, a);
  PG_CHECK_LAUNCH();
  return PG_OK;
}

template int launch_gru_fwd<float>(const GruFwdArgs<float>&, hipStream_t);
template int launch_gru_fwd<bf16_t>(const GruFwdArgs<bf16_t>&, hipStream_t);
template int launch_gru_bwd<float>(const GruBwdArgs<float>&, hipStream_t);
template int launch_gru_bwd<bf16_t>(const GruBwdArgs<bf16_t>&, hipStream_t);
