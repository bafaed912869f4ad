// VAE.sample (vae_model.py:300-358): autoregressive decoding, one launch per GRU cell per step plus one fused
// vocab-projection + softmax + draw kernel per step.
#include <algorithm>
#include <math.h>
#include "pg_gemm.h"
#include "pg_gru.h"
#include "pg_layout.h"
#include "pg_misc.h"
#include "pg_mma.h"
#include "pg_persist.h"
#include <type_traits>

template <typename T> static inline T* at(void* base, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }
template <typename T> static inline const T* at(const void* base, size_t off) {
  return reinterpret_cast<const T*>(static_cast<const char*>(base) + off);
}

__global__ void sample_init_kernel(int n, int max_len, int bos, int pad, uint8_t* __restrict__ tokens, int* __restrict__ end_pads,
                                   int* __restrict__ tok, int* __restrict__ eos) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long)n * max_len) tokens[i] = (i % max_len == 0) ? (uint8_t)bos : (uint8_t)pad;
  if (i < n) { end_pads[i] = max_len; tok[i] = bos; eos[i] = 0; }
}

// logits = h_top . W_fc^T + b ; p = softmax(logits / temp) ; w = inverse-CDF draw (fp32 running sum in vocab order,
// first k with cdf_k > u, clamped to V-1) or argmax ; then the eos bookkeeping of vae_model.py:342-351.
template <class Cfg>
__global__ __launch_bounds__(Cfg::NTHREADS) void sample_head_kernel(const typename Cfg::T* __restrict__ htop, int n, int H,
                                                                     const typename Cfg::T* __restrict__ fcw,
                                                                     const float* __restrict__ fcb, int V, float inv_temp,
                                                                     const float* __restrict__ uniforms, unsigned long long seed,
                                                                     long row_offset, int step, int multinomial, int eos_id,
                                                                     int max_len, uint8_t* __restrict__ tokens,
                                                                     int* __restrict__ end_pads, int* __restrict__ tok,
                                                                     int* __restrict__ eos, const uint8_t* __restrict__ forced,
                                                                     float* __restrict__ probs_out) {
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
  RowSrc<T> as{htop, H, m0, n};
  RowSrc<T> bs{fcw, H, 0, Cfg::BN};
  mma_mainloop_nt<Cfg, Cfg::TN, 0>(acc, as, bs, H, smem);
  constexpr int LDY = Cfg::BN + 1;             // BN = the padded vocabulary width (64 or 128)
  float* sl = reinterpret_cast<float*>(smem);  // [BM][LDY]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
#pragma unroll
  for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < Cfg::TN; ++tn)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        sl[Cfg::row_of(wm, tm, reg, lane) * LDY + Cfg::col_in_group(wn, tn, lane & 31)] = acc[tm][tn][reg];
  __syncthreads();
  const int row = threadIdx.x;
  const int b = m0 + row;
  if (row >= Cfg::BM || b >= n) return;
  float* y = sl + row * LDY;
  float mx = -INFINITY;
  int amax = 0;
  for (int v = 0; v < V; ++v) {
    const float x = (y[v] + fcb[v]) * inv_temp;
    y[v] = x;
    if (x > mx) { mx = x; amax = v; }
  }
  int w = amax;
  if (multinomial) {
    float s = 0.0f;
    for (int v = 0; v < V; ++v) {
      const float e = __expf(y[v] - mx);
      y[v] = e;
      s += e;
    }
    float u;
    if (uniforms) u = uniforms[(long)(step - 1) * n + b];
    else {
      const unsigned long long gr = (unsigned long long)(row_offset + b);
      Philox4 r = philox4x32_10((uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)step, 0x5A17u, (uint32_t)seed, (uint32_t)(seed >> 32));
      u = pg_u01(r.x);
    }
    if (probs_out) {   // parity hook: the probabilities the draw sees
      float* po = probs_out + ((long)(step - 1) * n + b) * Cfg::BN;
      for (int v = 0; v < Cfg::BN; ++v) po[v] = v < V ? y[v] / s : 0.0f;
    }
    float cdf = 0.0f;
    w = V - 1;
    for (int v = 0; v < V; ++v) {
      cdf += y[v] / s;
      if (cdf > u) { w = v; break; }
    }
  }
  tok[b] = forced ? (int)forced[(long)b * max_len + step] : w;   // parity hook: teacher forcing
  if (!eos[b]) {
    tokens[(long)b * max_len + step] = (uint8_t)w;
    if (w == eos_id) { end_pads[b] = step + 1; eos[b] = 1; }
  }
}

extern "C" size_t pg_sample_workspace_bytes(const pg_dims* d, int dtype, int n) {
  if (check_dims(d, dtype) != PG_OK || n < 1) return 0;
  return sample_ws_layout(d, dtype, n).total;
}

int check_persist();   // pg_vae.hip
int shadow_join(hipStream_t st);
static int sample_check_persist() { return check_persist(); }

template <typename T>
static int sample_t(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes, int n,
                    int max_len, const float* z, const float* uniforms, uint64_t seed, long row_offset, float temp,
                    int multinomial, uint8_t* tokens, int32_t* end_pads, const uint8_t* forced, float* probs_out,
                    hipStream_t st) {
  const ParamOff po = param_offsets(d);
  const ShadowOff so = shadow_offsets(d, dtype);
  const SampleWs w = sample_ws_layout(d, dtype, n);
  PG_REQUIRE(ws_bytes >= w.total, "sample: workspace too small (%zu < %zu)", ws_bytes, w.total);
  PG_REQUIRE(max_len >= 1, "sample: max_len must be >= 1");
  PG_REQUIRE(temp > 0.0f, "sample: temperature must be positive");
  const int Hd = d->Hd, Dz = d->Dz, L = d->L;
  hipLaunchKernelGGL(sample_init_kernel, dim3(cdiv((long)n * max_len, 256)), dim3(256), 0, st, n, max_len, d->bos, d->pad, tokens,
                     end_pads, at<int>(ws, w.tok), at<int>(ws, w.eos));
  PG_CHECK_LAUNCH();
  T* z_op = at<T>(ws, w.zop);
  hipLaunchKernelGGL(convert_kernel<T>, dim3(std::min<int>(cdiv((long)n * Dz, 256), 4096)), dim3(256), 0, st, z, z_op, (long)n * Dz);
  PG_CHECK_LAUNCH();
  {
    GemmEpi<T> e;
    e.C32 = at<float>(ws, w.zproj); e.ldc = 3 * Hd;
    PG_TRY(launch_gemm_nt<T>(z_op, Dz, at<T>(shadow, so.wih0z), Dz, n, 3 * Hd, Dz, e, st));
  }
  {
    GemmEpi<T> e;
    e.bias = params + po.lat_b; e.C32 = at<float>(ws, w.h0_32); e.Ct = at<T>(ws, w.h0_op); e.ldc = Hd;
    PG_TRY(launch_gemm_nt<T>(z_op, Dz, at<T>(shadow, so.lat_w), Dz, n, Hd, Dz, e, st));
  }
  for (int l = 0; l < L; ++l) {
    PG_CHECK_HIP(hipMemcpyAsync(at<float>(ws, w.carry[l]), at<float>(ws, w.h0_32), (size_t)n * Hd * 4, hipMemcpyDeviceToDevice, st));
    PG_CHECK_HIP(hipMemcpyAsync(at<T>(ws, w.hop[l][0]), at<T>(ws, w.h0_op), (size_t)n * Hd * sizeof(T), hipMemcpyDeviceToDevice, st));
  }
  // the whole decode loop in one group-persistent launch where it is covered (bf16, default decoder, at most 16384
  // rows, a 256-CU device); PG_NO_PERSIST bit 3 / pg_persist_disable force the per-step path below.  Measured faster at
  // every row count (n = 256: 2.2 vs 6.3 ms, n = 8192: 10.8 vs 21.0 ms per 100-token batch)
  if constexpr (std::is_same<T, bf16_t>::value) {
    static const long min_rows = getenv("PG_SAMPLE_PERSIST_MIN") ? atol(getenv("PG_SAMPLE_PERSIST_MIN")) : 1;
    constexpr long CHUNK = 16384;   // rows per launch (dec_sample_persist_supported); larger batches run chunk after chunk
    if (!(persist_off_mask() & 8) && n >= min_rows && max_len >= 2 && dec_sample_persist_supported(Hd, L, d->V, n < CHUNK ? n : CHUNK)) {
      for (long r0 = 0; r0 < n; r0 += CHUNK) {
        DecSamplePersist pa;
        pa.L = L; pa.n = (int)(n - r0 < CHUNK ? n - r0 : CHUNK); pa.T = max_len - 1;
        for (int l = 0; l < L; ++l) {
          pa.Whh[l] = at<T>(shadow, so.whh[l]);
          pa.bhn[l] = params + po.dec_bhh[l] + 2 * Hd;
          if (l > 0) { pa.Wih[l] = at<T>(shadow, so.wih[l]); pa.bias_gi[l] = at<float>(shadow, so.bias_gi[l]); }
          pa.hop[l][0] = at<T>(ws, w.hop[l][0]) + r0 * Hd; pa.hop[l][1] = at<T>(ws, w.hop[l][1]) + r0 * Hd;
          pa.carry[l] = at<float>(ws, w.carry[l]) + r0 * Hd;
        }
        pa.table = at<float>(shadow, so.tab_dec); pa.zproj = at<float>(ws, w.zproj) + r0 * 3 * Hd;
        pa.tok = at<int>(ws, w.tok) + r0; pa.eos = at<int>(ws, w.eos) + r0;
        pa.fcw = at<T>(shadow, so.fc_w); pa.fcb = at<float>(shadow, so.fc_b);
        pa.V = d->V; pa.multinomial = multinomial; pa.eos_id = d->eos; pa.max_len = max_len;
        pa.inv_temp = 1.0f / temp; pa.seed = seed; pa.row_offset = row_offset + r0;
        pa.uniforms = uniforms ? uniforms + r0 : nullptr; pa.u_stride = n;
        pa.tokens = tokens + r0 * max_len; pa.end_pads = end_pads + r0;
        pa.forced = forced ? forced + r0 * max_len : nullptr;
        const bool no_compact = getenv("PG_SAMPLE_NO_COMPACT") != nullptr;   // A/B aid, read per call
        pa.compact = (forced || no_compact) ? 0 : 1;
        pa.probs_out = probs_out ? probs_out + r0 * 64 : nullptr; pa.p_stride = n;   // (V <= 64 here: dec_sample_persist_supported)
        pa.sync = at<unsigned>(ws, w.sync);
        PG_TRY(launch_dec_sample_persist(pa, st));
      }
      return PG_OK;
    }
  }
  using HCfg = TileCfg<T, 128, 64, 8, 4, 1>;
  using HCfg2 = TileCfg<T, 128, 128, 8, 4, 1>;   // V in (64, 128]
  static_assert(HCfg::SMEM_BYTES >= 128 * 65 * 4 && HCfg2::SMEM_BYTES >= 128 * 129 * 4, "head kernel LDS too small for the logits tile");
  const bool wide = vpad(d) == 128;
  for (int i = 1; i < max_len; ++i) {
    const int p = (i - 1) & 1;
    for (int l = 0; l < L; ++l) {
      GruFwdArgs<T> a;
      a.rows = n; a.H = Hd;
      a.hprev_op = at<T>(ws, w.hop[l][p]);
      a.hprev32 = at<float>(ws, w.carry[l]);
      a.Whh = at<T>(shadow, so.whh[l]);
      a.bhn = params + po.dec_bhh[l] + 2 * Hd;
      if (l == 0) {
        a.gi_mode = PG_GI_TABLE;
        a.table = at<float>(shadow, so.tab_dec);
        a.tok = at<int>(ws, w.tok);
        a.zproj = at<float>(ws, w.zproj);
      } else {
        a.gi_mode = PG_GI_FUSED;
        a.x = at<T>(ws, w.hop[l - 1][p ^ 1]);
        a.Kx = Hd;
        a.Wih = at<T>(shadow, so.wih[l]);
        a.bias_gi = at<float>(shadow, so.bias_gi[l]);
      }
      a.h_op = at<T>(ws, w.hop[l][p ^ 1]);
      a.h32 = at<float>(ws, w.carry[l]);
      PG_TRY(launch_gru_fwd<T>(a, st));
    }
    if (!wide)
      PG_LAUNCH_BIG_SMEM(sample_head_kernel<HCfg>, dim3(cdiv(n, HCfg::BM)), dim3(HCfg::NTHREADS), HCfg::SMEM_BYTES, st,
                         at<T>(ws, w.hop[L - 1][p ^ 1]), n, Hd, at<T>(shadow, so.fc_w), at<float>(shadow, so.fc_b), d->V,
                         1.0f / temp, uniforms, (unsigned long long)seed, row_offset, i, multinomial, d->eos, max_len, tokens,
                         end_pads, at<int>(ws, w.tok), at<int>(ws, w.eos), forced, probs_out);
    else
      PG_LAUNCH_BIG_SMEM(sample_head_kernel<HCfg2>, dim3(cdiv(n, HCfg2::BM)), dim3(HCfg2::NTHREADS), HCfg2::SMEM_BYTES, st,
                         at<T>(ws, w.hop[L - 1][p ^ 1]), n, Hd, at<T>(shadow, so.fc_w), at<float>(shadow, so.fc_b), d->V,
                         1.0f / temp, uniforms, (unsigned long long)seed, row_offset, i, multinomial, d->eos, max_len, tokens,
                         end_pads, at<int>(ws, w.tok), at<int>(ws, w.eos), forced, probs_out);
    PG_CHECK_LAUNCH();
  }
  return PG_OK;
}

extern "C" int pg_sample_forced(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes,
                                int n, int max_len, const float* z, const float* uniforms, uint64_t seed, long row_offset,
                                float temp, int multinomial, uint8_t* tokens, int32_t* end_pads, const uint8_t* forced,
                                float* probs_out, void* stream) {
  PG_TRY(check_dims(d, dtype));
  PG_REQUIRE(params && shadow && ws && z && tokens && end_pads, "pg_sample: NULL buffer");
  PG_REQUIRE(n >= 1, "pg_sample: n must be >= 1");
  PG_REQUIRE(!probs_out || multinomial, "pg_sample_forced: probabilities are only computed when multinomial != 0");
  PG_TRY(sample_check_persist());
  PG_TRY(shadow_join((hipStream_t)stream));
  return dtype == PG_BF16 ? sample_t<bf16_t>(d, dtype, params, shadow, ws, ws_bytes, n, max_len, z, uniforms, seed, row_offset,
                                             temp, multinomial, tokens, end_pads, forced, probs_out, (hipStream_t)stream)
                          : sample_t<float>(d, dtype, params, shadow, ws, ws_bytes, n, max_len, z, uniforms, seed, row_offset,
                                            temp, multinomial, tokens, end_pads, forced, probs_out, (hipStream_t)stream);
}

extern "C" int pg_sample(const pg_dims* d, int dtype, const float* params, const void* shadow, void* ws, size_t ws_bytes, int n,
                         int max_len, const float* z, const float* uniforms, uint64_t seed, long row_offset, float temp,
                         int multinomial, uint8_t* tokens, int32_t* end_pads, void* stream) {
  return pg_sample_forced(d, dtype, params, shadow, ws, ws_bytes, n, max_len, z, uniforms, seed, row_offset, temp, multinomial,
                          tokens, end_pads, nullptr, nullptr, stream);
}
