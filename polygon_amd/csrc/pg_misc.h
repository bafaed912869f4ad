// Elementwise / reduction / small kernels of the VAE hot path (HBM- or latency-bound work around the MFMA kernels).
#pragma once
#include "pg_common.h"
#include "pg_layout.h"
#include "pg_scatter.h"   // the embedding-table gradient (scatter-sum by token)

// ---- wave / block reductions (64-wide waves) ----------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- shadow weights ------------------------------------------------------------------------------
struct CopyDesc {
  long src_off;    // element offset into params (fp32)
  size_t dst_off;  // byte offset into shadow
  int rows, cols;  // source sub-matrix
  int src_ld, dst_ld;
  int transpose;   // dst[c*dst_ld + r] instead of dst[r*dst_ld + c]
  int to_f32;      // destination is fp32 whatever T is
};
#define PG_MAX_COPY 48
struct CopyTable {
  int n;
  CopyDesc d[PG_MAX_COPY];
};

template <typename T> __global__ void copy_table_kernel(const float* __restrict__ params, char* __restrict__ shadow, CopyTable tab) {
  const CopyDesc d = tab.d[blockIdx.y];
  const long total = (long)d.rows * d.cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // iterate in DESTINATION order so that stores coalesce
    int r, c;
    if (d.transpose) { c = (int)(i / d.rows); r = (int)(i % d.rows); }
    else { r = (int)(i / d.cols); c = (int)(i % d.cols); }
    const float v = params[d.src_off + (long)r * d.src_ld + c];
    const long o = d.transpose ? (long)c * d.dst_ld + r : (long)r * d.dst_ld + c;
    if (d.to_f32) reinterpret_cast<float*>(shadow + d.dst_off)[o] = v;
    else reinterpret_cast<T*>(shadow + d.dst_off)[o] = from_f32<T>(v);
  }
}

// tab[v, g] = sum_e emb[v,e] * Wih[g, e] + b_ih[g] + (g < 2H ? b_hh[g] : 0)
// Block = 64 gate columns x a slice of the vocabulary: the 64 W_ih rows (read with consecutive threads on consecutive
// elements of a row) and the embedding rows go through LDS once; a thread then owns one column and walks its vocabulary
// rows with the W row in a conflict-free padded LDS line.  (First version: one thread per output with both operands read
// from global memory in a 51-step dependent chain, 16 us on the step's critical path in front of the encoder forward.)
#define PG_GI_TAB_COLS 64
#define PG_GI_TAB_VB 16
static __global__ __launch_bounds__(256) void gi_table_kernel(const float* __restrict__ emb, int E, const float* __restrict__ Wih,
                                                               int ldw, const float* __restrict__ bih, const float* __restrict__ bhh,
                                                               int H, int V, float* __restrict__ tab) {
  extern __shared__ float gts[];             // [64][LD] W rows (LD odd: conflict-free columns), then [PG_GI_TAB_VB][E] embedding rows
  const int LD = E | 1;
  float* sW = gts;
  float* sE = gts + PG_GI_TAB_COLS * LD;
  const int g0 = blockIdx.x * PG_GI_TAB_COLS, v0 = blockIdx.y * PG_GI_TAB_VB;
  for (int i = threadIdx.x; i < PG_GI_TAB_COLS * E; i += blockDim.x) {
    const int r = i / E, k = i % E;
    sW[r * LD + k] = g0 + r < 3 * H ? Wih[(long)(g0 + r) * ldw + k] : 0.0f;
  }
  for (int i = threadIdx.x; i < PG_GI_TAB_VB * E; i += blockDim.x) {
    const int r = i / E, k = i % E;
    sE[i] = v0 + r < V ? emb[(long)(v0 + r) * E + k] : 0.0f;
  }
  __syncthreads();
  const int c = threadIdx.x & 63, vq = threadIdx.x >> 6;     // column, quarter of the vocabulary slice
  const int g = g0 + c;
  if (g >= 3 * H) return;
  const float bias = bih[g] + (g < 2 * H ? bhh[g] : 0.0f);
  const float* w = sW + c * LD;
#pragma unroll
  for (int j = 0; j < PG_GI_TAB_VB / 4; ++j) {
    const int vl = vq * (PG_GI_TAB_VB / 4) + j;
    if (v0 + vl >= V) break;
    const float* e = sE + vl * E;
    float s0 = bias, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int k = 0;
    for (; k + 4 <= E; k += 4) {
      s0 = fmaf(e[k], w[k], s0); s1 = fmaf(e[k + 1], w[k + 1], s1);
      s2 = fmaf(e[k + 2], w[k + 2], s2); s3 = fmaf(e[k + 3], w[k + 3], s3);
    }
    for (; k < E; ++k) s0 = fmaf(e[k], w[k], s0);
    tab[(long)(v0 + vl) * 3 * H + g] = (s0 + s1) + (s2 + s3);
  }
}
static inline int launch_gi_table(const float* emb, int E, const float* Wih, int ldw, const float* bih, const float* bhh, int H,
                                  int V, float* tab, hipStream_t st) {
  const size_t smem = ((size_t)PG_GI_TAB_COLS * (E | 1) + (size_t)PG_GI_TAB_VB * E) * sizeof(float);   // E <= 256: <= 82 KB
  PG_LAUNCH_BIG_SMEM(gi_table_kernel, dim3((3 * H + PG_GI_TAB_COLS - 1) / PG_GI_TAB_COLS, (V + PG_GI_TAB_VB - 1) / PG_GI_TAB_VB),
                     dim3(256), smem, st, emb, E, Wih, ldw, bih, bhh, H, V, tab);
  PG_CHECK_LAUNCH();
  return PG_OK;
}
static __global__ void bias_fold_kernel(const float* __restrict__ bih, const float* __restrict__ bhh, int H, float* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < 3 * H) out[g] = bih[g] + (g < 2 * H ? bhh[g] : 0.0f);
}

// ---- packed batch ----------------------------------------------------------------------------------
// mode 0: encoder (tok only).  mode 1: decoder (tok, target = next token, t, b) + offsets copy.
// zero (optional): n_zero words the pass must find cleared (loss sums, tickets, counters) - written here instead of by a
// memset of its own in front of the recurrence
template <typename T>
static __global__ void build_packed_kernel(const int32_t* __restrict__ tokens, int B, int Tmax, int pad, Sched s, int mode,
                                    int* __restrict__ tok, int* __restrict__ tgt, int* __restrict__ rt, int* __restrict__ rb,
                                    int* __restrict__ off_out, unsigned* __restrict__ zero, int n_zero) {
  const int t = blockIdx.x;
  if (off_out && blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = threadIdx.x; i <= s.T + 1; i += blockDim.x) off_out[i] = s.off[i];
  if (zero && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0)   // (the last step's block has the fewest rows to pack)
    for (int i = threadIdx.x; i < n_zero; i += blockDim.x) zero[i] = 0u;
  const int Bt = s.off[t + 1] - s.off[t];
  const int b = blockIdx.y * blockDim.x + threadIdx.x;
  if (b >= Bt) return;
  const int r = s.off[t] + b;
  const int v = tokens[(long)b * Tmax + t];
  tok[r] = v;
  if (mode == 1) {
    tgt[r] = (t + 1 < Tmax) ? tokens[(long)b * Tmax + t + 1] : pad;
    rt[r] = t;
    rb[r] = b;
  }
}

template <typename T> __global__ void convert_kernel(const float* __restrict__ in, T* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = from_f32<T>(in[i]);
}

// ---- encoder head: reparameterisation + KL (vae_model.py:232-236) ---------------------------------
// The block whose ticket is the last one finalises the loss (kl = sum / B_global -> scal[SC_KL], *kl_out): no one-thread
// kernel behind this one on the step's critical path.  A block's sum has been performed when its atomic has RETURNED
// (the returned value is consumed before the ticket is drawn), so the last ticket holder reads the complete sum.
template <typename T>
static __global__ void reparam_kl_kernel(const float* __restrict__ mulv, const float* __restrict__ eps, int B, int Dz,
                                  float* __restrict__ z, T* __restrict__ z_op, float* __restrict__ mu_out,
                                  float* __restrict__ lv_out, float* __restrict__ eps_save, float* __restrict__ kl_sum,
                                  float inv_B, float* __restrict__ kl_out) {
  // grid-stride over the B x Dz elements: few blocks, because every block ends in two atomics on ONE address (~12 ns each,
  // serialised at L2: 512 blocks of 256 elements spent 12 us there)
  float term = 0.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)B * Dz; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / Dz), k = (int)(i % Dz);
    const float mu = mulv[(long)b * 2 * Dz + k], lv = mulv[(long)b * 2 * Dz + Dz + k];
    const float e = eps ? eps[i] : 0.0f;
    const float zz = mu + __expf(0.5f * lv) * e;
    z[i] = zz;
    if (z_op) z_op[i] = from_f32<T>(zz);
    if (mu_out) mu_out[i] = mu;
    if (lv_out) lv_out[i] = lv;
    if (eps_save) eps_save[i] = e;
    term += 0.5f * (__expf(lv) + mu * mu - 1.0f - lv);
  }
  term = wave_sum(term);
  __shared__ float part[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) part[wv] = term;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += part[w];
    const float before = atomicAdd(kl_sum, s);
    unsigned* ticket = reinterpret_cast<unsigned*>(kl_sum + 1);      // SC_KL: zeroed with SC_KL_SUM, overwritten below
    asm volatile("" ::"v"(before) : "memory");                       // the returned value is waited for: the sum is in place
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      const float kl = atomicAdd(kl_sum, 0.0f) * inv_B;
      kl_sum[1] = kl;
      if (kl_out) *kl_out = kl;
    }
  }
}

// dq[b, 0:Dz] = dmu, dq[b, Dz:2Dz] = dlogvar
template <typename T>
static __global__ void reparam_bwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps, const float* __restrict__ dz,
                                   float s_host, const float* __restrict__ g_dev, int B, int Dz, float* __restrict__ dq,
                                   T* __restrict__ dq_op) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Dz) return;
  const int b = (int)(i / Dz), k = (int)(i % Dz);
  const float s_kl = s_host * (g_dev ? *g_dev : 1.0f);   // upstream gradient of kl x 1 / B_global
  const float mu = mulv[(long)b * 2 * Dz + k], lv = mulv[(long)b * 2 * Dz + Dz + k];
  const float g = dz ? dz[i] : 0.0f;
  const float dmu = g + s_kl * mu;
  const float dlv = g * eps[i] * 0.5f * __expf(0.5f * lv) + s_kl * 0.5f * (__expf(lv) - 1.0f);
  const long o = (long)b * 2 * Dz + k;
  dq[o] = dmu; dq[o + Dz] = dlv;
  dq_op[o] = from_f32<T>(dmu); dq_op[o + Dz] = from_f32<T>(dlv);
}

// ---- cross entropy: one wave per packed row; lane k holds vocabulary entries k and (NV = 2: Vp = 128) k + 64 -----------
template <typename T, int NV>
static __global__ void ce_kernel(const float* __restrict__ logits, const int* __restrict__ tgt, long N, int V, int Vp, int pad,
                          T* __restrict__ dlogits, float* __restrict__ scal) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  float loss = 0.0f, cnt = 0.0f;
  // grid-stride over rows, one wave per row: few blocks => few atomics on the two loss words
  for (long r = (long)blockIdx.x * nw + wv; r < N; r += (long)gridDim.x * nw) {
    float x[NV], e[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = lane + 64 * i;
      x[i] = v < V ? logits[r * Vp + v] : -INFINITY;
      mx = fmaxf(mx, x[i]);
    }
    mx = wave_max(mx);
    float se = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      e[i] = lane + 64 * i < V ? __expf(x[i] - mx) : 0.0f;
      se += e[i];
    }
    const float s = wave_sum(se);
    const int tg = tgt[r];
    const bool w = tg != pad;
    float xt = __shfl(x[0], tg & 63, 64);
    if (NV > 1) {
      const float x1 = __shfl(x[NV - 1], tg & 63, 64);
      xt = tg >= 64 ? x1 : xt;
    }
    if (w) { loss += __logf(s) + mx - xt; cnt += 1.0f; }
    if (dlogits) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        if (v < Vp) dlogits[r * Vp + v] = from_f32<T>(w && v < V ? e[i] / s - (v == tg ? 1.0f : 0.0f) : 0.0f);
      }
    }
  }
  __shared__ float pl[16], pc[16];
  if (lane == 0) { pl[wv] = loss; pc[wv] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, c = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += pl[w]; c += pc[w]; }
    atomicAdd(scal + SC_CE_SUM, a);
    atomicAdd(scal + SC_CE_CNT, c);
  }
}

// the block that draws the last CE ticket finalises (fused kernel); the plain path keeps the one-thread kernel below
__device__ __forceinline__ void ce_add_and_finalize(float* scal, float loss, float cnt, unsigned nblocks, float count_override,
                                                    float* recon_out) {
  const float b0 = atomicAdd(scal + SC_CE_SUM, loss);
  const float b1 = atomicAdd(scal + SC_CE_CNT, cnt);
  asm volatile("" ::"v"(b0), "v"(b1) : "memory");                      // the returned values are waited for: both sums are in place
  if (atomicAdd(reinterpret_cast<unsigned*>(scal + SC_CE_TICKET), 1u) == nblocks - 1) {
    const float sum = atomicAdd(scal + SC_CE_SUM, 0.0f);
    const float c = count_override > 0.0f ? count_override : atomicAdd(scal + SC_CE_CNT, 0.0f);
    scal[SC_INV_CNT] = 1.0f / c;
    scal[SC_RECON] = sum / c;
    if (recon_out) *recon_out = sum / c;
  }
}
static __global__ void finalize_ce_kernel(float* scal, float count_override, float* recon_out) {
  const float cnt = count_override > 0.0f ? count_override : scal[SC_CE_CNT];
  scal[SC_INV_CNT] = 1.0f / cnt;
  scal[SC_RECON] = scal[SC_CE_SUM] / cnt;
  if (recon_out) *recon_out = scal[SC_RECON];
}
// slot = g_host * (*g_dev) * (which ? scal[which] : mul)
static __global__ void grad_scale_kernel(float* scal, int slot, float g_host, const float* g_dev, int src_slot, float mul) {
  scal[slot] = g_host * (g_dev ? *g_dev : 1.0f) * (src_slot >= 0 ? scal[src_slot] : mul);
}

// ---- column sums (bias gradients): out[c] += alpha * sum_r X[r, c] --------------------------------
template <typename T>
static __global__ void colsum_kernel(const T* __restrict__ X, long R, int C, long ld, long rows_per_block, float* __restrict__ out,
                              float alpha, const float* __restrict__ alpha_dev) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float s = 0.0f;
  if (c < C)
    for (long r = r0 + ry; r < r1; r += 4) s += to_f32<T>(X[r * ld + c]);
  __shared__ float part[4][64];
  part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < C) {
    const float a = alpha * (alpha_dev ? *alpha_dev : 1.0f);
    atomicAdd(out + c, a * (part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]));
  }
}

// dzproj[b, c] = sum_{t : b < B_t} X[off[t] + b, c]     (gradient of the per-sequence z projection)
// One thread owns 8 consecutive columns of one sequence (16-byte loads; the 64 threads of a wave cover 512 contiguous
// columns of a row, a workgroup 4 sequences), walks that sequence's steps four at a time so that four independent loads
// are in flight per thread, and sums in fp32.  HBM-bound: reads X once (first version: 2-byte loads, one dependent load
// per step, 110 us for 154 MB; this one streams).
template <typename T>
static __global__ __launch_bounds__(256) void rowseg_sum_kernel(const T* __restrict__ X, const int* __restrict__ off, int Tn, int C,
                                                                 int B, float* __restrict__ out, T* __restrict__ out_op,
                                                                 float* __restrict__ zero, long n_zero) {
  constexpr int EPC = 16 / sizeof(T);
  __shared__ int soff[PG_MAX_T + 2];
  for (int i = threadIdx.x; i <= Tn; i += blockDim.x) soff[i] = off[i];
  // (zero, optional: a buffer the NEXT kernels of the chain accumulate into - cleared here instead of by a memset between them)
  if (zero)
    for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n_zero;
         i += (long)gridDim.x * gridDim.y * blockDim.x)
      zero[i] = 0.0f;
  __syncthreads();
  const int b = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * EPC;
  if (b >= B || c >= C) return;
  int len = 0;                                   // steps this sequence takes part in: B_t = off[t+1] - off[t] is non-increasing
  {
    int lo = 0, hi = Tn;                         // first t with B_t <= b
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (soff[mid + 1] - soff[mid] > b) lo = mid + 1; else hi = mid;
    }
    len = lo;
  }
  float s[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s[e] = 0.0f;
  auto add = [&](const uint4& v) __attribute__((always_inline)) {
    const T* x = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] += to_f32<T>(x[e]);
  };
  int t = 0;
  for (; t + 8 <= len; t += 8) {                 // (8 loads in flight per thread: the chain of ~60 steps is latency-bound)
    uint4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint4*>(X + (long)(soff[t + i] + b) * C + c);
#pragma unroll
    for (int i = 0; i < 8; ++i) add(v[i]);
  }
  for (; t + 4 <= len; t += 4) {
    uint4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const uint4*>(X + (long)(soff[t + i] + b) * C + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) add(v[i]);
  }
  for (; t < len; ++t) add(*reinterpret_cast<const uint4*>(X + (long)(soff[t] + b) * C + c));
  float* o = out + (long)b * C + c;
  T* oo = out_op + (long)b * C + c;
#pragma unroll
  for (int e = 0; e < EPC; ++e) { o[e] = s[e]; oo[e] = from_f32<T>(s[e]); }
}

// C[m*ldc + n] += sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]   (all fp32; the V-row table gradients -> x_emb / W_ih0)
// One 16 x 16 output tile and one 64-wide slice of K per block, operands staged in LDS with the loads laid along whichever
// stride is 1, atomic accumulation (split K; and x_emb's gradient is accumulated by two streams at once, pg_vae.hip).
// (Was one wave per output element: 46 k waves of 30 products each for W_ih0's gradient, 40 us beside the large GEMMs.)
#define PG_SG_KC 64
static __global__ __launch_bounds__(256) void small_gemm_kernel(const float* __restrict__ A, long sam, long sak,
                                                                const float* __restrict__ Bm, long sbk, long sbn, int M, int N,
                                                                int K, float* __restrict__ C, long ldc, int skip_row) {
  __shared__ float sA[16][PG_SG_KC + 1];
  __shared__ float sB[PG_SG_KC][17];
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16, k0 = blockIdx.z * PG_SG_KC;
  for (int i = threadIdx.x; i < 16 * PG_SG_KC; i += 256) {
    const int m = sak == 1 ? i / PG_SG_KC : i % 16, k = sak == 1 ? i % PG_SG_KC : i / 16;
    sA[m][k] = (m0 + m < M && k0 + k < K) ? A[(m0 + m) * sam + (k0 + k) * sak] : 0.0f;
  }
  for (int i = threadIdx.x; i < 16 * PG_SG_KC; i += 256) {
    const int n = sbn == 1 ? i % 16 : i / PG_SG_KC, k = sbn == 1 ? i / 16 : i % PG_SG_KC;
    sB[k][n] = (n0 + n < N && k0 + k < K) ? Bm[(k0 + k) * sbk + (n0 + n) * sbn] : 0.0f;
  }
  __syncthreads();
  const int tm = threadIdx.x >> 4, tn = threadIdx.x & 15;
  float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 8
  for (int k = 0; k < PG_SG_KC; k += 2) {
    s0 = fmaf(sA[tm][k], sB[k][tn], s0);
    s1 = fmaf(sA[tm][k + 1], sB[k + 1][tn], s1);
  }
  const int m = m0 + tm, n = n0 + tn;
  if (m < M && n < N && m != skip_row) atomicAdd(C + m * ldc + n, s0 + s1);
}

// ---- optimiser: global-norm clip + Adam over the flat buffers -----------------------------------------
#define PG_SUMSQ_BLOCKS 512
static __global__ void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  // 16-byte loads, two independent sums per thread (the scalar version read 18 MB at 1.1 TB/s: 16 us behind the join)
  float s = 0.0f, s2 = 0.0f;
  const long n4 = ((reinterpret_cast<size_t>(g) & 15) == 0) ? n / 4 : 0;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = g4[i];
    s = fmaf(v.x, v.x, s); s2 = fmaf(v.y, v.y, s2); s = fmaf(v.z, v.z, s); s2 = fmaf(v.w, v.w, s2);
  }
  for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s = fmaf(g[i], g[i], s);
  s = wave_sum(s + s2);
  __shared__ float part[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) part[wv] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += part[w];
    partial[blockIdx.x] = a;
  }
}
// zero_g (pg_clip_adam_zero): the gradient buffer is left zeroed, on both paths, so that the next step need not fill it
static __global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, const float* __restrict__ partial, int npartial, float max_norm, float step_size,
                            float beta1, float beta2, float inv_sqrt_bc2, float eps, float* __restrict__ gnorm_out,
                            const unsigned* __restrict__ fail, float* __restrict__ zero_g) {
  // a persistent recurrence kernel of this step gave up (sticky device word): its gradients are garbage, so the update is
  // skipped here, on the device, with no host round trip; the host reports the failure at its next check
  if (fail && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    if (zero_g)
      for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) zero_g[i] = 0.0f;
    return;
  }
  __shared__ float red[256];
  float s = 0.0f;
  for (int i = threadIdx.x; i < npartial; i += blockDim.x) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float total = sqrtf(red[0]);
  if (gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) *gnorm_out = total;
  float coef = 1.0f;
  if (max_norm > 0.0f) coef = fminf(max_norm / (total + 1e-6f), 1.0f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    if (zero_g) zero_g[i] = 0.0f;
  }
}
