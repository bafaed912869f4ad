// GRU recurrent step kernels for gfx950: gate GEMM on MFMA + fused gate (or gate-gradient) epilogue.
#include <stdlib.h>
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
  if (a.hprev_op && !(a.dbg & 1)) {
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
  if (a.dbg & 2) { if (acc[0][0][0] == 123.456f) a.h32[0] = 1.0f; return; }
  // Epilogue in two phases per 16-row group: FIRST gather every input into registers (independent loads, all in flight
  // together), THEN compute and store.  Interleaving loads and stores per element would serialise one memory round
  // trip per element, because the output pointers may alias the inputs as far as the compiler knows.
#pragma unroll
  for (int tu = 0; tu < TU; ++tu) {
    const int j = j0 + Cfg::col_in_group(wn, tu, lane & 31);
    const bool jok = j < H;
    const float bhn = jok ? a.bhn[j] : 0.0f;
    float bg_r = 0.f, bg_z = 0.f, bg_n = 0.f;
    if (MODE == PG_GI_FUSED && jok) {
      bg_r = a.bias_gi[j]; bg_z = a.bias_gi[H + j]; bg_n = a.bias_gi[2 * H + j];
    }
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm) {
      float gr[16], gz[16], gn[16], hp[16];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int b = m0 + Cfg::row_of(wm, tm, reg, lane);
        const bool ok = jok && b < a.rows;
        gr[reg] = bg_r; gz[reg] = bg_z; gn[reg] = bg_n; hp[reg] = 0.0f;
        if (ok) {
          if (MODE == PG_GI_ARRAY) {
            const float* g = a.gi + (long)b * 3 * H;
            gr[reg] = g[j]; gz[reg] = g[H + j]; gn[reg] = g[2 * H + j];
          } else if (MODE == PG_GI_TABLE) {
            const float* g = a.table + (long)a.tok[b] * 3 * H;
            gr[reg] = g[j]; gz[reg] = g[H + j]; gn[reg] = g[2 * H + j];
            if (a.zproj) {
              const float* zp = a.zproj + (long)b * 3 * H;
              gr[reg] += zp[j]; gz[reg] += zp[H + j]; gn[reg] += zp[2 * H + j];
            }
          }
          if (a.hprev32) hp[reg] = a.hprev32[(long)b * H + j];
        }
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int b = m0 + Cfg::row_of(wm, tm, reg, lane);
        const bool ok = jok && b < a.rows;
        float gnn = gn[reg];
        if (MODE == PG_GI_FUSED) gnn += acc[tm][3 * TU + tu][reg];
        const float r = sigmoidf_(gr[reg] + acc[tm][tu][reg]);
        const float z = sigmoidf_(gz[reg] + acc[tm][TU + tu][reg]);
        const float hn = acc[tm][2 * TU + tu][reg] + bhn;
        const float n = tanhf_(gnn + r * hn);
        const float h = (1.0f - z) * n + z * hp[reg];
        if (ok) {
          const long o = (long)b * H + j;
          a.h32[o] = h;
          a.h_op[o] = from_f32<T>(h);
          if (a.sv_r) {
            a.sv_r[o] = from_f32<T>(r);
            a.sv_z[o] = from_f32<T>(z);
            a.sv_n[o] = from_f32<T>(n);
            a.sv_hn[o] = from_f32<T>(hn);
            a.sv_hprev[o] = from_f32<T>(hp[reg]);
          }
          if (a.h_drop)
            a.h_drop[o] = from_f32<T>(h * dropout_scale(a.seed, a.layer, a.t, b * a.b_mul + a.b_add, j, a.p_drop, inv_keep));
        }
      }
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
  if (a.rows_in > 0 && m0 < a.rows_in && !(a.dbg & 1)) {
    RowSrc<T> as{a.dGH_in, 3L * H, m0, a.rows_in};
    RowSrc<T> bs{a.WhhT, 3L * H, j0, H};
    mma_mainloop_nt<Cfg, Cfg::TN, 0>(acc, as, bs, 3 * H, smem);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  if (a.dbg & 2) { if (acc[0][0][0] == 123.456f) a.dhz_out[0] = 1.0f; return; }
#pragma unroll
  for (int tn = 0; tn < Cfg::TN; ++tn) {
    const int j = j0 + Cfg::col_in_group(wn, tn, lane & 31);
    const bool jok = j < H;
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
      for (int half = 0; half < 2; ++half) {   // 8 rows at a time: gather all inputs first, then compute + store
        float dh[8], r[8], z[8], n[8], hn[8], hp[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int reg = half * 8 + q;
          const int b = m0 + Cfg::row_of(wm, tm, reg, lane);
          const bool ok = jok && b < a.rows_out;
          const long o = (long)b * H + j;
          float d = 0.0f;
          r[q] = z[q] = n[q] = hn[q] = hp[q] = 0.0f;
          if (ok) {
            if (b < a.rows_in) d = acc[tm][tn][reg] + a.dhz_in[o];
            else if (a.inject) d = a.inject[o];
            if (!a.final_step) {
              if (a.dout) d += a.dout[o];
              r[q] = to_f32<T>(a.sv_r[o]); z[q] = to_f32<T>(a.sv_z[o]); n[q] = to_f32<T>(a.sv_n[o]);
              hn[q] = to_f32<T>(a.sv_hn[o]); hp[q] = to_f32<T>(a.sv_hprev[o]);
            } else if (a.dh0_accumulate) {
              d += a.dh0[o];
            }
          }
          dh[q] = d;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int reg = half * 8 + q;
          const int b = m0 + Cfg::row_of(wm, tm, reg, lane);
          const bool ok = jok && b < a.rows_out;
          if (!ok) continue;
          const long o = (long)b * H + j;
          if (a.final_step) {
            a.dh0[o] = dh[q];
            continue;
          }
          const float dnp = dh[q] * (1.0f - z[q]) * (1.0f - n[q] * n[q]);
          const float dzp = dh[q] * (hp[q] - n[q]) * z[q] * (1.0f - z[q]);
          const float drp = dnp * hn[q] * r[q] * (1.0f - r[q]);
          const long og = (long)b * 3 * H + j;
          a.dGI[og] = from_f32<T>(drp);
          a.dGI[og + H] = from_f32<T>(dzp);
          a.dGI[og + 2 * H] = from_f32<T>(dnp);
          a.dGH[og] = from_f32<T>(drp);
          a.dGH[og + H] = from_f32<T>(dzp);
          a.dGH[og + 2 * H] = from_f32<T>(dnp * r[q]);
          a.dhz_out[o] = dh[q] * z[q];
        }
      }
  }
}

// ================================================================================================
// v2 step kernels for the training regime (rows <= a few thousand): latency-optimised.
//   * block = G K-groups x 4 waves; every group contracts 1/G of K into its own LDS staging region, all groups' loads
//     are in flight together (the step is latency-bound: few blocks, short K loop);
//   * accumulators are exchanged / reduced through LDS as fp32 tiles, then EVERY thread runs the gate epilogue on
//     8 consecutive columns of one row with 16-byte loads and stores (instead of 2-byte lane-per-column accesses);
//   * the epilogue's inputs are fetched before the main loop, so their latency hides under it.
// ================================================================================================
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = x;
  }
};
__device__ __forceinline__ void zero8(float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.0f;
}

constexpr int V2_BM = 64, V2_BH = 64, V2_LD = 68;   // exchange tile row stride in floats (272 B: 16-B aligned rows)

// writes / accumulates the wave's 32x32 accumulator tiles into the fp32 exchange tiles xt[slot][64][V2_LD]
template <class Cfg, int NACC>
__device__ __forceinline__ void v2_exchange(f32x16 (&acc)[Cfg::TM][NACC], float* xt, int gtid, int group, int G) {
  const int lane = gtid & 63, wave = gtid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  for (int g = 0; g < G; ++g) {
    if (group == g) {
#pragma unroll
      for (int s = 0; s < NACC; ++s)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          float* p = xt + (s * V2_BM + Cfg::row_of(wm, 0, reg, lane)) * V2_LD + wn * 32 + (lane & 31);
          if (g == 0) *p = acc[0][s][reg];
          else *p += acc[0][s][reg];
        }
    }
    __syncthreads();
  }
}

template <typename T, int MODE_T, int G, int D>
__device__ __forceinline__ void gru_fwd2_body(const GruFwdArgs<T>& a, int bx, int by, char* smem) {
  using Cfg = TileCfg<T, V2_BM, 3 * V2_BH, 8, 2, 2, 3, D>;
  static_assert(Cfg::TM == 1 && Cfg::TU == 1, "v2 geometry");
  constexpr int NACC = (MODE_T == PG_GI_FUSED || MODE_T == PG_GI_DYN) ? 4 : 3;
  constexpr int NTH = 256 * G;
  const int MODE = MODE_T == PG_GI_DYN ? a.gi_mode : MODE_T;   // block-uniform
  const int H = a.H;
  const int j0 = bx * V2_BH, m0 = by * V2_BM;
  const int tid = threadIdx.x, gtid = tid & 255;
  const int group = __builtin_amdgcn_readfirstlane(tid >> 8);   // wave-uniform, and provably so (scalar descriptors/offsets)
  // ---- epilogue assignment + input prefetch: chunk c covers row c/8, columns 8*(c%8)..+7 of the 64x64 tile
  constexpr int CH = (V2_BM * V2_BH / 8 + NTH - 1) / NTH;
  float gr[CH][8], gz[CH][8], gn[CH][8], hp[CH][8];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int id = tid + c * NTH;
    const int b = m0 + id / 8, j = j0 + (id % 8) * 8;
    zero8(gr[c]); zero8(gz[c]); zero8(gn[c]); zero8(hp[c]);
    if (id < V2_BM * V2_BH / 8 && b < a.rows && j < H && !(a.dbg & 4)) {
      if (MODE == PG_GI_ARRAY) {
        const float* g = a.gi + (long)b * 3 * H + j;
        Vec8<float>::load(g, gr[c]); Vec8<float>::load(g + H, gz[c]); Vec8<float>::load(g + 2 * H, gn[c]);
      } else if (MODE == PG_GI_TABLE) {
        const float* g = a.table + (long)a.tok[b] * 3 * H + j;
        Vec8<float>::load(g, gr[c]); Vec8<float>::load(g + H, gz[c]); Vec8<float>::load(g + 2 * H, gn[c]);
        if (a.zproj) {
          float t0[8], t1[8], t2[8];
          const float* zp = a.zproj + (long)b * 3 * H + j;
          Vec8<float>::load(zp, t0); Vec8<float>::load(zp + H, t1); Vec8<float>::load(zp + 2 * H, t2);
#pragma unroll
          for (int i = 0; i < 8; ++i) { gr[c][i] += t0[i]; gz[c][i] += t1[i]; gn[c][i] += t2[i]; }
        }
      } else {
        Vec8<float>::load(a.bias_gi + j, gr[c]); Vec8<float>::load(a.bias_gi + H + j, gz[c]);
        Vec8<float>::load(a.bias_gi + 2 * H + j, gn[c]);
      }
      if (a.hprev32) Vec8<float>::load(a.hprev32 + (long)b * H + j, hp[c]);
    }
  }
  // ---- gate GEMM, K split over the G groups
  f32x16 acc[1][NACC];
  {
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NACC; ++s) acc[0][s] = zacc;
  }
  char* gsm = smem + group * Cfg::SMEM_BYTES;
  {
    // one pipelined K loop over [h_{t-1} | x] against [W_hh | W_ih]
    const int K1 = (a.hprev_op && !(a.dbg & 1)) ? H : 0;
    const int K2 = (NACC == 4 && MODE == PG_GI_FUSED) ? a.Kx : 0;
    const int nkt = (K1 + Cfg::BK - 1) / Cfg::BK + (K2 + Cfg::BK - 1) / Cfg::BK;
    const int nkg = (nkt + G - 1) / G;
    if (nkt > 0) {
      RowSrc<T> a1{K1 > 0 ? a.hprev_op : a.x, H, m0, a.rows};
      GateRowSrc<T> b1{K1 > 0 ? a.Whh : a.Wih, H, H, j0, V2_BH};
      RowSrc<T> a2{a.x, a.Kx, m0, a.rows};
      GateRowSrc<T> b2{a.Wih, a.Kx, H, j0, V2_BH};
      mma_mainloop_nt2<Cfg, NACC, 2, NACC == 4 ? 3 : 2>(acc, a1, b1, K1, a2, b2, K2, gsm, gtid, group * nkg, nkg);
    }
  }
  float* xt = reinterpret_cast<float*>(smem);
  if (a.dbg & 2) { if (acc[0][0][0] == 123.456f || gr[0][0] == 77.7f) a.h32[0] = 1.0f; return; }
  v2_exchange<Cfg, NACC>(acc, xt, gtid, group, G);
  if (a.dbg & 8) { if (xt[tid] == 123.456f) a.h32[0] = 1.0f; return; }
  // ---- wide epilogue
  const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int id = tid + c * NTH;
    const int row = id / 8, col = (id % 8) * 8;
    const int b = m0 + row, j = j0 + col;
    if (!(id < V2_BM * V2_BH / 8 && b < a.rows && j < H)) continue;
    float ar[8], az[8], ahn[8], ain[8], bhn[8];
    Vec8<float>::load(xt + (0 * V2_BM + row) * V2_LD + col, ar);
    Vec8<float>::load(xt + (1 * V2_BM + row) * V2_LD + col, az);
    Vec8<float>::load(xt + (2 * V2_BM + row) * V2_LD + col, ahn);
    if (NACC == 4) Vec8<float>::load(xt + (3 * V2_BM + row) * V2_LD + col, ain);
    Vec8<float>::load(a.bhn + j, bhn);
    float h[8], r[8], z[8], n[8], hn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      r[i] = sigmoidf_(gr[c][i] + ar[i]);
      z[i] = sigmoidf_(gz[c][i] + az[i]);
      hn[i] = ahn[i] + bhn[i];
      const float gnn = (NACC == 4 && MODE == PG_GI_FUSED) ? gn[c][i] + ain[i] : gn[c][i];
      n[i] = tanhf_(gnn + r[i] * hn[i]);
      h[i] = (1.0f - z[i]) * n[i] + z[i] * hp[c][i];
    }
    const long o = (long)b * H + j;
    Vec8<float>::store(a.h32 + o, h);
    Vec8<T>::store(a.h_op + o, h);
    if (a.sv_r) {
      Vec8<T>::store(a.sv_r + o, r); Vec8<T>::store(a.sv_z + o, z); Vec8<T>::store(a.sv_n + o, n);
      Vec8<T>::store(a.sv_hn + o, hn); Vec8<T>::store(a.sv_hprev + o, hp[c]);
    }
    if (a.h_drop) {
      float hd[8], m0[4], m1[4];
      dropout_scale4(a.seed, a.layer, a.t, b * a.b_mul + a.b_add, j >> 2, a.p_drop, inv_keep, m0);
      dropout_scale4(a.seed, a.layer, a.t, b * a.b_mul + a.b_add, (j >> 2) + 1, a.p_drop, inv_keep, m1);
#pragma unroll
      for (int i = 0; i < 4; ++i) { hd[i] = h[i] * m0[i]; hd[4 + i] = h[4 + i] * m1[i]; }
      Vec8<T>::store(a.h_drop + o, hd);
    }
  }
}

template <typename T, int MODE, int G, int D>
__global__ __launch_bounds__(256 * G) void gru_fwd2_kernel(GruFwdArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gru_fwd2_body<T, MODE, G, D>(a, blockIdx.x, blockIdx.y, smem);
}
template <typename T, int G, int D>
__global__ __launch_bounds__(256 * G) void gru_fwd2_diag_kernel(GruFwdDiag<T> d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GruFwdArgs<T>& a = d.a[blockIdx.z];
  if ((int)blockIdx.y * V2_BM >= a.rows) return;   // block-uniform: this cell has fewer rows than the widest one
  gru_fwd2_body<T, PG_GI_DYN, G, D>(a, blockIdx.x, blockIdx.y, smem);
}

template <typename T, int G, int D, bool UP, bool PRE = true>
__device__ __forceinline__ void gru_bwd2_body(const GruBwdArgs<T>& a, int bx, int by, char* smem) {
  // PRE: fetch the epilogue inputs before the K loop (latency hidden, +48 VGPRs) or after it (register-tight builds)
  using Cfg = TileCfg<T, V2_BM, V2_BH, 8, 2, 2, 1, D>;
  constexpr int NTH = 256 * G;
  constexpr int NACC = UP ? 2 : 1;
  const int H = a.H;
  const int j0 = bx * V2_BH, m0 = by * V2_BM;
  const int tid = threadIdx.x, gtid = tid & 255;
  const int group = __builtin_amdgcn_readfirstlane(tid >> 8);   // wave-uniform, and provably so (scalar descriptors/offsets)
  constexpr int CH = (V2_BM * V2_BH / 8 + NTH - 1) / NTH;
  float dh[CH][8], r[CH][8], z[CH][8], n[CH][8], hn[CH][8], hp[CH][8];
  auto gather = [&]() __attribute__((always_inline)) {
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int id = tid + c * NTH;
    const int b = m0 + id / 8, j = j0 + (id % 8) * 8;
    zero8(dh[c]); zero8(r[c]); zero8(z[c]); zero8(n[c]); zero8(hn[c]); zero8(hp[c]);
    if (id < V2_BM * V2_BH / 8 && b < a.rows_out && j < H && !(a.dbg & 4)) {
      const long o = (long)b * H + j;
      if (b < a.rows_in) Vec8<float>::load(a.dhz_in + o, dh[c]);
      else if (a.inject) Vec8<float>::load(a.inject + o, dh[c]);
      float t[8];
      if (!a.final_step) {
        if (a.dout) {
          Vec8<float>::load(a.dout + o, t);
#pragma unroll
          for (int i = 0; i < 8; ++i) dh[c][i] += t[i];
        }
        Vec8<T>::load(a.sv_r + o, r[c]); Vec8<T>::load(a.sv_z + o, z[c]); Vec8<T>::load(a.sv_n + o, n[c]);
        Vec8<T>::load(a.sv_hn + o, hn[c]); Vec8<T>::load(a.sv_hprev + o, hp[c]);
      } else if (a.dh0_accumulate) {
        Vec8<float>::load(a.dh0 + o, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) dh[c][i] += t[i];
      }
    }
  }
  };
  if (PRE) gather();
  f32x16 acc[1][NACC];
  {
    const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NACC; ++s) acc[0][s] = zacc;
  }
  {
    // one pipelined K loop over [dGH_{t+1} | dGI_up] against [W_hh^T | W_ih_up^T]
    const int K1 = (a.rows_in > 0 && m0 < a.rows_in && !(a.dbg & 1)) ? 3 * H : 0;   // block-uniform
    const int K2 = (UP && a.dGI_up && !a.final_step) ? 3 * H : 0;
    const int nkt = (K1 + Cfg::BK - 1) / Cfg::BK + (K2 + Cfg::BK - 1) / Cfg::BK;
    const int nkg = (nkt + G - 1) / G;
    if (nkt > 0) {
      RowSrc<T> a1{a.dGH_in, 3L * H, m0, a.rows_in};
      RowSrc<T> b1{a.WhhT, 3L * H, j0, H};
      RowSrc<T> a2{a.dGI_up, 3L * H, m0, a.rows_out};
      RowSrc<T> b2{a.WihT_up, 3L * H, j0, H};
      mma_mainloop_nt2<Cfg, NACC, 0, NACC - 1>(acc, a1, b1, K1, a2, b2, K2, smem + group * Cfg::SMEM_BYTES, gtid,
                                               group * nkg, nkg);
    }
  }
  float* xt = reinterpret_cast<float*>(smem);
  if (a.dbg & 2) { if (acc[0][0][0] == 123.456f || dh[0][0] == 77.7f) a.dhz_out[0] = 1.0f; return; }
  v2_exchange<Cfg, NACC>(acc, xt, gtid, group, G);
  if (a.dbg & 8) { if (xt[tid] == 123.456f) a.dhz_out[0] = 1.0f; return; }
  if (!PRE) gather();
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int id = tid + c * NTH;
    const int row = id / 8, col = (id % 8) * 8;
    const int b = m0 + row, j = j0 + col;
    if (!(id < V2_BM * V2_BH / 8 && b < a.rows_out && j < H)) continue;
    float ac[8];
    Vec8<float>::load(xt + row * V2_LD + col, ac);
    const long o = (long)b * H + j;
    if (b < a.rows_in) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dh[c][i] += ac[i];
    }
    if constexpr (UP) {
      if (a.dGI_up && !a.final_step) {
        float au[8];
        Vec8<float>::load(xt + (V2_BM + row) * V2_LD + col, au);
        const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
        float m0[4] = {1.0f, 1.0f, 1.0f, 1.0f}, m1[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (a.p_drop > 0.0f) {
          dropout_scale4(a.seed, a.layer, a.t, b * a.b_mul + a.b_add, j >> 2, a.p_drop, inv_keep, m0);
          dropout_scale4(a.seed, a.layer, a.t, b * a.b_mul + a.b_add, (j >> 2) + 1, a.p_drop, inv_keep, m1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { dh[c][i] += m0[i] * au[i]; dh[c][4 + i] += m1[i] * au[4 + i]; }
      }
    }
    if (a.final_step) {
      Vec8<float>::store(a.dh0 + o, dh[c]);
      continue;
    }
    float drp[8], dzp[8], dnp[8], dnr[8], dhz[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dnp[i] = dh[c][i] * (1.0f - z[c][i]) * (1.0f - n[c][i] * n[c][i]);
      dzp[i] = dh[c][i] * (hp[c][i] - n[c][i]) * z[c][i] * (1.0f - z[c][i]);
      drp[i] = dnp[i] * hn[c][i] * r[c][i] * (1.0f - r[c][i]);
      dnr[i] = dnp[i] * r[c][i];
      dhz[i] = dh[c][i] * z[c][i];
    }
    const long og = (long)b * 3 * H + j;
    Vec8<T>::store(a.dGI + og, drp); Vec8<T>::store(a.dGI + og + H, dzp); Vec8<T>::store(a.dGI + og + 2 * H, dnp);
    Vec8<T>::store(a.dGH + og, drp); Vec8<T>::store(a.dGH + og + H, dzp); Vec8<T>::store(a.dGH + og + 2 * H, dnr);
    Vec8<float>::store(a.dhz_out + o, dhz);
  }
}

template <typename T, int G, int D>
__global__ __launch_bounds__(256 * G) void gru_bwd2_kernel(GruBwdArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gru_bwd2_body<T, G, D, false, (G < 4)>(a, blockIdx.x, blockIdx.y, smem);
}
template <typename T, int G, int D>
__global__ __launch_bounds__(256 * G) void gru_bwd2_diag_kernel(GruBwdDiag<T> d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GruBwdArgs<T>& a = d.a[blockIdx.z];
  if ((int)blockIdx.y * V2_BM >= a.rows_out) return;
  gru_bwd2_body<T, G, D, true, (G < 4)>(a, blockIdx.x, blockIdx.y, smem);
}

template <int MODE, int G, int D, typename T> static int launch_fwd2(const GruFwdArgs<T>& a, hipStream_t stream) {
  using Cfg = TileCfg<T, V2_BM, 3 * V2_BH, 8, 2, 2, 3, D>;
  constexpr size_t smem = (size_t)G * Cfg::SMEM_BYTES;
  static_assert(smem >= 4 * V2_BM * V2_LD * 4, "exchange tiles must fit the staging LDS");
  dim3 grid(cdiv(a.H, V2_BH), cdiv(a.rows, V2_BM));
  PG_LAUNCH_BIG_SMEM((gru_fwd2_kernel<T, MODE, G, D>), grid, dim3(256 * G), smem, stream, a);
  PG_CHECK_LAUNCH();
  return PG_OK;
}
template <int G, int D, typename T> static int launch_bwd2(const GruBwdArgs<T>& a, hipStream_t stream) {
  using Cfg = TileCfg<T, V2_BM, V2_BH, 8, 2, 2, 1, D>;
  constexpr size_t smem = (size_t)G * Cfg::SMEM_BYTES;
  static_assert(smem >= V2_BM * V2_LD * 4, "exchange tile must fit the staging LDS");
  dim3 grid(cdiv(a.H, V2_BH), cdiv(a.rows_out, V2_BM));
  PG_LAUNCH_BIG_SMEM((gru_bwd2_kernel<T, G, D>), grid, dim3(256 * G), smem, stream, a);
  PG_CHECK_LAUNCH();
  return PG_OK;
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
  static const int variant = getenv("PG_FWD_VARIANT") ? atoi(getenv("PG_FWD_VARIANT")) : 0;   // tuning aid
  if (variant == 1 || variant == 2) {
    if (variant == 1) {
      if (a.gi_mode == PG_GI_ARRAY) return launch_fwd2<PG_GI_ARRAY, 1, 2, T>(a, stream);
      if (a.gi_mode == PG_GI_TABLE) return launch_fwd2<PG_GI_TABLE, 1, 2, T>(a, stream);
      return launch_fwd2<PG_GI_FUSED, 1, 2, T>(a, stream);
    }
    if (a.gi_mode == PG_GI_ARRAY) return launch_fwd2<PG_GI_ARRAY, 2, 2, T>(a, stream);
    if (a.gi_mode == PG_GI_TABLE) return launch_fwd2<PG_GI_TABLE, 2, 2, T>(a, stream);
    return launch_fwd2<PG_GI_FUSED, 2, 2, T>(a, stream);
  }
  if (variant == 3) {
    using Cfg = TileCfg<T, 128, 192, 4, 2, 2, 3, 2>;
    if (a.gi_mode == PG_GI_ARRAY) return launch_fwd_cfg<Cfg, PG_GI_ARRAY, T>(a, stream);
    if (a.gi_mode == PG_GI_TABLE) return launch_fwd_cfg<Cfg, PG_GI_TABLE, T>(a, stream);
    return launch_fwd_cfg<Cfg, PG_GI_FUSED, T>(a, stream);
  }
  if (variant == 4) {
    using Cfg = TileCfg<T, 128, 192, 8, 2, 2, 3, 1>;
    if (a.gi_mode == PG_GI_ARRAY) return launch_fwd_cfg<Cfg, PG_GI_ARRAY, T>(a, stream);
    if (a.gi_mode == PG_GI_TABLE) return launch_fwd_cfg<Cfg, PG_GI_TABLE, T>(a, stream);
    return launch_fwd_cfg<Cfg, PG_GI_FUSED, T>(a, stream);
  }
  PG_REQUIRE(a.H % 8 == 0, "gru_fwd: hidden size must be a multiple of 8");
  if (a.rows > 2048) {   // throughput regime (sampling): one K-group, two blocks per CU
    if (a.gi_mode == PG_GI_ARRAY) return launch_fwd2<PG_GI_ARRAY, 1, 2, T>(a, stream);
    if (a.gi_mode == PG_GI_TABLE) return launch_fwd2<PG_GI_TABLE, 1, 2, T>(a, stream);
    return launch_fwd2<PG_GI_FUSED, 1, 2, T>(a, stream);
  }
  if (a.gi_mode == PG_GI_ARRAY) return launch_fwd2<PG_GI_ARRAY, 2, 2, T>(a, stream);
  if (a.gi_mode == PG_GI_TABLE) return launch_fwd2<PG_GI_TABLE, 2, 2, T>(a, stream);
  return launch_fwd2<PG_GI_FUSED, 2, 1, T>(a, stream);
}

template <typename T> int launch_gru_bwd(const GruBwdArgs<T>& a, hipStream_t stream) {
  constexpr int EPC = 16 / sizeof(T);
  if (a.rows_out <= 0) return PG_OK;
  PG_REQUIRE(a.H % EPC == 0, "gru_bwd: hidden size %d must be a multiple of %d", a.H, EPC);
  PG_REQUIRE(a.rows_in <= a.rows_out, "gru_bwd: rows_in %d > rows_out %d (batch must be sorted by length)", a.rows_in, a.rows_out);
  if (a.rows_out <= 4096) return launch_bwd2<4, 1, T>(a, stream);
  using Cfg = TileCfg<T, 64, 64, 8, 2, 2, 1, 4>;
  dim3 grid(cdiv(a.H, Cfg::BN), cdiv(a.rows_out, Cfg::BM));
  PG_LAUNCH_BIG_SMEM(gru_bwd_kernel<Cfg>, grid, dim3(Cfg::NTHREADS), Cfg::SMEM_BYTES, stream, a);
  PG_CHECK_LAUNCH();
  return PG_OK;
}

template <typename T> int launch_gru_fwd_diag(const GruFwdDiag<T>& d, hipStream_t stream) {
  if (d.n <= 0) return PG_OK;
  constexpr int G = 2, D = 1;
  using Cfg = TileCfg<T, V2_BM, 3 * V2_BH, 8, 2, 2, 3, D>;
  constexpr size_t smem = (size_t)G * Cfg::SMEM_BYTES;
  int rows = 0, H = d.a[0].H;
  for (int i = 0; i < d.n; ++i) {
    PG_REQUIRE(d.a[i].H == H && H % 8 == 0, "gru_fwd_diag: all cells must share one hidden size (multiple of 8)");
    PG_REQUIRE(d.a[i].gi_mode != PG_GI_FUSED || d.a[i].Kx % (16 / (int)sizeof(T)) == 0, "gru_fwd_diag: bad fused input width");
    rows = rows > d.a[i].rows ? rows : d.a[i].rows;
  }
  if (rows <= 0) return PG_OK;
  dim3 grid(cdiv(H, V2_BH), cdiv(rows, V2_BM), d.n);
  PG_LAUNCH_BIG_SMEM((gru_fwd2_diag_kernel<T, G, D>), grid, dim3(256 * G), smem, stream, d);
  PG_CHECK_LAUNCH();
  return PG_OK;
}
template <typename T> int launch_gru_bwd_diag(const GruBwdDiag<T>& d, hipStream_t stream) {
  if (d.n <= 0) return PG_OK;
  constexpr int G = 4, D = 1;
  using Cfg = TileCfg<T, V2_BM, V2_BH, 8, 2, 2, 1, D>;
  constexpr size_t smem = (size_t)G * Cfg::SMEM_BYTES;
  static_assert(smem >= 2 * V2_BM * V2_LD * 4, "exchange tiles must fit the staging LDS");
  int rows = 0, H = d.a[0].H;
  for (int i = 0; i < d.n; ++i) {
    PG_REQUIRE(d.a[i].H == H && H % 8 == 0, "gru_bwd_diag: all cells must share one hidden size (multiple of 8)");
    PG_REQUIRE(d.a[i].rows_in <= d.a[i].rows_out, "gru_bwd_diag: rows_in > rows_out (batch must be sorted by length)");
    rows = rows > d.a[i].rows_out ? rows : d.a[i].rows_out;
  }
  if (rows <= 0) return PG_OK;
  dim3 grid(cdiv(H, V2_BH), cdiv(rows, V2_BM), d.n);
  PG_LAUNCH_BIG_SMEM((gru_bwd2_diag_kernel<T, G, D>), grid, dim3(256 * G), smem, stream, d);
  PG_CHECK_LAUNCH();
  return PG_OK;
}
template int launch_gru_fwd_diag<float>(const GruFwdDiag<float>&, hipStream_t);
template int launch_gru_fwd_diag<bf16_t>(const GruFwdDiag<bf16_t>&, hipStream_t);
template int launch_gru_bwd_diag<float>(const GruBwdDiag<float>&, hipStream_t);
template int launch_gru_bwd_diag<bf16_t>(const GruBwdDiag<bf16_t>&, hipStream_t);

template int launch_gru_fwd<float>(const GruFwdArgs<float>&, hipStream_t);
template int launch_gru_fwd<bf16_t>(const GruFwdArgs<bf16_t>&, hipStream_t);
template int launch_gru_bwd<float>(const GruBwdArgs<float>&, hipStream_t);
template int launch_gru_bwd<bf16_t>(const GruBwdArgs<bf16_t>&, hipStream_t);
