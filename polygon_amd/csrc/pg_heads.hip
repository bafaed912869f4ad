// The two row-local stretches between the recurrences of a training step, each as ONE kernel (bf16 operands):
//
//   enc_heads_kernel:  [mu | logvar] = h_fin . [W_mu ; W_lv]^T + b,  z = mu + exp(logvar / 2) . eps,  KL   (vae_model.py:231-236)
//   dec_zin_kernel:    z -> bf16,  zproj = z . W_ih0[:, E:]^T,  h0 = z . W_lat^T + b_lat                     (vae_model.py:253-258)
//
// As separate launches (a 16-block GEMM + the element-wise kernel; a convert + two GEMMs of 96 / 32 blocks) they were
// 28 + 35 us of launch- and latency-bound stream time on the step's critical path, with the chip idle: a 128 x 128-tile
// GEMM with K = 128..256 is one short dependent chain per block.  Here a block owns 16 ROWS and all the outputs: the rows
// sit in LDS (bf16), every wave streams the weight rows of its output columns straight from L2 as MFMA fragments (the
// load pattern of the persistent kernels' weight prologue: lane (lr, lq) reads 16 bytes of row lr at K chunk lq), all of a
// tile's fragments requested before its first MFMA.  v_mfma_f32_16x16x32_bf16 with the weights as first operand: the
// accumulator tile is [output column][row], lane (lr, lq) holds columns 4lq..4lq+3 of row lr -> 16-byte stores.
#include <hip/hip_runtime.h>
#include "pg_host.h"
#include "pg_layout.h"
#include "pg_misc.h"
#include "pg_mma.h"

namespace {

constexpr int HR = 16;   // rows per block

__device__ __forceinline__ f32x4 mma16h(const bf16x8& w, const bf16x8& x, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
}

// grid = (ceil(B / 16), Y), block = 256 (4 waves).  Block y owns the mu columns [y * Dz / Y, (y + 1) * Dz / Y) and the logvar
// columns of the same latent dimensions (waves 0, 1: mu; waves 2, 3: logvar), so that z and the KL terms of its
// dimensions are block-local; Y = 2 halves each wave's chain of weight-fragment loads and doubles the blocks in flight.
// LDS: [16][K + 8] bf16 rows, then [16][2 * Dz / Y + 4] fp32 results.  Requirements (host-checked): K % 32 == 0,
// Dz / Y % 32 == 0.
template <int NT>   // 16-column tiles per wave = Dz / Y / 32
__global__ __launch_bounds__(256) void enc_heads_kernel(const bf16_t* __restrict__ hfin, int B, int K, const bf16_t* __restrict__ wq,
                                                         const float* __restrict__ bq, int Dz, const float* __restrict__ eps,
                                                         float* __restrict__ mulv, float* __restrict__ z, bf16_t* __restrict__ z_op,
                                                         float* __restrict__ mu_out, float* __restrict__ lv_out,
                                                         float* __restrict__ eps_save, float* __restrict__ kl_sum, float inv_B,
                                                         float* __restrict__ kl_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = 2 * Dz, ROWB = K * 2 + 16;
  const int DB = Dz / gridDim.y, d0 = blockIdx.y * DB;     // this block's latent dimensions
  const int LDM = 2 * DB + 4;                              // sM row: [mu of the DB dimensions | logvar of them]
  char* sA = smem;
  float* sM = reinterpret_cast<float*>(smem + HR * ROWB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * HR;
  for (int c = tid; c < HR * (K / 8); c += 256) {
    const int r = c / (K / 8), ch = c % (K / 8);
    const int b = m0 + r < B ? m0 + r : B - 1;
    *reinterpret_cast<uint4*>(sA + r * ROWB + ch * 16) = *reinterpret_cast<const uint4*>(hfin + (long)b * K + ch * 8);
  }
  __syncthreads();
  const int half = wave >> 1;                              // 0: mu rows of W, 1: logvar rows
  const int lc0 = (wave & 1) * (DB / 2);                   // first of this wave's columns within the block's DB
  const int col0 = half * Dz + d0 + lc0;                   // ... as a row of [W_mu ; W_lv] / a column of mulv
  const int KS = K / 32;
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* wrow = wq + (long)(col0 + nt * 16 + lr) * K + lq * 8;
    const char* xrow = sA + lr * ROWB + lq * 16;
    for (int ks = 0; ks < KS; ks += 4) {   // K % 128 == 0 is the common case; the tail steps are guarded
      bf16x8 wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ks + j < KS) wv[j] = *reinterpret_cast<const bf16x8*>(wrow + (ks + j) * 32);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ks + j < KS) acc[nt] = mma16h(wv[j], *reinterpret_cast<const bf16x8*>(xrow + (ks + j) * 64), acc[nt]);
    }
  }
  const int b = m0 + lr;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = col0 + nt * 16 + lq * 4;
    const float4 bias = *reinterpret_cast<const float4*>(bq + c);
    const float4 v = make_float4(acc[nt][0] + bias.x, acc[nt][1] + bias.y, acc[nt][2] + bias.z, acc[nt][3] + bias.w);
    *reinterpret_cast<float4*>(sM + lr * LDM + half * DB + lc0 + nt * 16 + lq * 4) = v;
    if (b < B) *reinterpret_cast<float4*>(mulv + (long)b * N + c) = v;
  }
  __syncthreads();
  float term = 0.0f;
  for (int i = tid; i < HR * DB; i += 256) {
    const int r = i / DB, k = i % DB;
    const int bb = m0 + r;
    if (bb >= B) continue;
    const float mu = sM[r * LDM + k], lv = sM[r * LDM + DB + k];
    const long o = (long)bb * Dz + d0 + k;
    const float e = eps ? eps[o] : 0.0f;
    const float zz = mu + __expf(0.5f * lv) * e;
    z[o] = zz;
    if (z_op) z_op[o] = (bf16_t)zz;
    if (mu_out) mu_out[o] = mu;
    if (lv_out) lv_out[o] = lv;
    if (eps_save) eps_save[o] = e;
    term += 0.5f * (__expf(lv) + mu * mu - 1.0f - lv);
  }
  term = wave_sum(term);
  __shared__ float part[4];
  if (lane == 0) part[wave] = term;
  __syncthreads();
  if (tid == 0) {   // the last block (ticket) finalises the loss: reparam_kl_kernel's protocol (pg_misc.h)
    const float before = atomicAdd(kl_sum, part[0] + part[1] + part[2] + part[3]);
    asm volatile("" ::"v"(before) : "memory");
    if (atomicAdd(reinterpret_cast<unsigned*>(kl_sum + 1), 1u) == gridDim.x * gridDim.y - 1) {
      const float kl = atomicAdd(kl_sum, 0.0f) * inv_B;
      kl_sum[1] = kl;
      if (kl_out) *kl_out = kl;
    }
  }
}

// grid = (ceil(B / 16), Y), block = 512 (8 waves; the 8 Y waves of a row tile split the N = 4 Hd columns
// [zproj (3 Hd) | h0 (Hd)] evenly; Y = 2 when that leaves whole pairs of 16-column tiles).  Requirements (host-checked): Dz % 32 == 0, Dz <= 256, Hd % 32 == 0.
template <int KS>   // Dz / 32
__global__ __launch_bounds__(512) void dec_zin_kernel(const float* __restrict__ z, int B, int Hd, const bf16_t* __restrict__ wih0z,
                                                       const bf16_t* __restrict__ latw, const float* __restrict__ latb,
                                                       bf16_t* __restrict__ z_op, float* __restrict__ zproj, float* __restrict__ h0_32,
                                                       bf16_t* __restrict__ h0_op) {
  constexpr int Dz = KS * 32, ROWB = Dz * 2 + 16;
  __shared__ __attribute__((aligned(16))) char sA[HR * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * HR;
  for (int i = tid; i < HR * (Dz / 4); i += 512) {   // 4 values per thread: fp32 in, bf16 to LDS and to z_op
    const int r = i / (Dz / 4), c4 = i % (Dz / 4);
    const int bb = m0 + r;
    const float4 v = bb < B ? *reinterpret_cast<const float4*>(z + (long)bb * Dz + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    bf16_t o[4] = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
    *reinterpret_cast<uint2*>(sA + r * ROWB + c4 * 8) = *reinterpret_cast<const uint2*>(o);
    if (bb < B && blockIdx.y == 0) *reinterpret_cast<uint2*>(z_op + (long)bb * Dz + c4 * 4) = *reinterpret_cast<const uint2*>(o);
  }
  __syncthreads();
  const int N = 4 * Hd, per_wave = N / (8 * gridDim.y), ntile = per_wave / 16;
  const int wcol0 = (blockIdx.y * 8 + wave) * per_wave;
  const int b = m0 + lr;
  bf16x8 x[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) x[ks] = *reinterpret_cast<const bf16x8*>(sA + lr * ROWB + (ks * 4 + lq) * 16);
  for (int nt = 0; nt < ntile; nt += 2) {   // two tiles' fragments in flight (ntile is even: per_wave % 32 == 0, host-checked)
    bf16x8 wv[2][KS];
    int col[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      col[j] = wcol0 + (nt + j) * 16;                    // tile-uniform: a tile lies in ONE of the two matrices (3 Hd % 16 == 0)
      const bf16_t* wrow = col[j] < 3 * Hd ? wih0z + (long)(col[j] + lr) * Dz : latw + (long)(col[j] - 3 * Hd + lr) * Dz;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wv[j][ks] = *reinterpret_cast<const bf16x8*>(wrow + (ks * 4 + lq) * 8);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = mma16h(wv[j][ks], x[ks], acc);
      if (b >= B) continue;
      const int c = col[j] + lq * 4;
      if (col[j] < 3 * Hd) {
        *reinterpret_cast<float4*>(zproj + (long)b * 3 * Hd + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      } else {
        const int ch = c - 3 * Hd;
        const float4 bias = *reinterpret_cast<const float4*>(latb + ch);
        const float4 v = make_float4(acc[0] + bias.x, acc[1] + bias.y, acc[2] + bias.z, acc[3] + bias.w);
        *reinterpret_cast<float4*>(h0_32 + (long)b * Hd + ch) = v;
        bf16_t o[4] = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
        *reinterpret_cast<uint2*>(h0_op + (long)b * Hd + ch) = *reinterpret_cast<const uint2*>(o);
      }
    }
  }
}

}  // namespace

bool heads_fused_enabled() {
  static const bool off = getenv("PG_NO_FUSED_HEADS") != nullptr;   // A/B aid: the separate GEMM + element-wise launches
  return !off;
}

// q heads + reparameterisation + KL of pg_encoder_fwd / pg_encoder_gen_fwd.  false: shape not covered, nothing launched.
bool launch_enc_heads(const bf16_t* hfin, int B, int K, const bf16_t* wq, const float* bq, int Dz, const float* eps, float* mulv,
                      float* z, bf16_t* z_op, float* mu_out, float* lv_out, float* eps_save, float* kl_sum, float inv_B,
                      float* kl_out, hipStream_t st, int* status) {
  *status = PG_OK;
  if (!heads_fused_enabled() || K % 32 != 0 || K > 1024 || Dz % 32 != 0) return false;
  const int Y = Dz % 64 == 0 ? 2 : 1, nt = Dz / Y / 32;       // tiles per wave
  if (nt != 1 && nt != 2 && nt != 4) return false;
  const size_t smem = (size_t)HR * (K * 2 + 16) + (size_t)HR * (2 * Dz / Y + 4) * 4;
  auto go = [&](auto kern) -> int {
    PG_LAUNCH_BIG_SMEM(kern, dim3(cdiv(B, HR), Y), dim3(256), smem, st, hfin, B, K, wq, bq, Dz, eps, mulv, z, z_op, mu_out, lv_out,
                       eps_save, kl_sum, inv_B, kl_out);
    PG_CHECK_LAUNCH();
    return PG_OK;
  };
  switch (nt) {
    case 1: *status = go(enc_heads_kernel<1>); break;
    case 2: *status = go(enc_heads_kernel<2>); break;
    default: *status = go(enc_heads_kernel<4>); break;
  }
  return true;
}

// z -> bf16, z projection of decoder layer 0, h0 of pg_decoder_fwd.  false: shape not covered, nothing launched.
bool launch_dec_zin(const float* z, int B, int Dz, int Hd, const bf16_t* wih0z, const bf16_t* latw, const float* latb, bf16_t* z_op,
                    float* zproj, float* h0_32, bf16_t* h0_op, hipStream_t st, int* status) {
  *status = PG_OK;
  if (!heads_fused_enabled() || Dz % 32 != 0 || Dz > 256 || Hd % 64 != 0) return false;
  auto go = [&](auto kern) -> int {
    hipLaunchKernelGGL(kern, dim3(cdiv(B, HR), Hd % 128 == 0 ? 2 : 1), dim3(512), 0, st, z, B, Hd, wih0z, latw, latb, z_op, zproj, h0_32, h0_op);
    PG_CHECK_LAUNCH();
    return PG_OK;
  };
  switch (Dz / 32) {
    case 1: *status = go(dec_zin_kernel<1>); break;
    case 2: *status = go(dec_zin_kernel<2>); break;
    case 4: *status = go(dec_zin_kernel<4>); break;
    case 8: *status = go(dec_zin_kernel<8>); break;
    default: return false;
  }
  return true;
}
