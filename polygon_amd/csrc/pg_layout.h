// Host-side layouts: flat parameter buffer, shadow weights, training / sampling workspaces, packed schedule.
#pragma once
#include "../../include/polygon_hip.h"
#include "pg_common.h"

struct ParamOff {
  long x_emb, enc_wih, enc_whh, enc_bih, enc_bhh, qmu_w, qmu_b, qlv_w, qlv_b, lat_w, lat_b;   // enc_* = layer 0, forward
  // every (layer, direction) of the encoder GRU, nn.GRU's parameter order (weight_ih_l{l}[_reverse], weight_hh, bias_ih, bias_hh)
  long g_wih[PG_MAX_ENC_LAYERS][2], g_whh[PG_MAX_ENC_LAYERS][2], g_bih[PG_MAX_ENC_LAYERS][2], g_bhh[PG_MAX_ENC_LAYERS][2];
  long dec_wih[PG_MAX_LAYERS], dec_whh[PG_MAX_LAYERS], dec_bih[PG_MAX_LAYERS], dec_bhh[PG_MAX_LAYERS];
  long fc_w, fc_b, total;
};

static inline int vpad(const pg_dims* d) { return PG_VPAD(d->V); }
// encoder shape (vae_model.py:63-81); 0 / 0 in pg_dims = one layer, one direction
static inline int enc_layers(const pg_dims* d) { return d->Lq > 0 ? d->Lq : 1; }
static inline int enc_dirs(const pg_dims* d) { return d->qbidir ? 2 : 1; }
static inline bool enc_generic(const pg_dims* d) { return enc_layers(d) > 1 || enc_dirs(d) > 1; }
static inline int enc_in(const pg_dims* d, int l) { return l == 0 ? d->E : d->Hq * enc_dirs(d); }

static inline int check_dims(const pg_dims* d, int dtype) {
  PG_REQUIRE(d != nullptr, "dims is NULL");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_BF16, "dtype %d is neither PG_F32 nor PG_BF16", dtype);
  PG_REQUIRE(d->V >= 2 && d->V <= PG_MAX_V, "vocabulary size %d outside [2,%d]", d->V, PG_MAX_V);
  PG_REQUIRE(d->E >= 1 && d->E <= 256, "embedding dim %d outside [1,256]", d->E);
  PG_REQUIRE(d->L >= 1 && d->L <= PG_MAX_LAYERS, "decoder layers %d outside [1,%d]", d->L, PG_MAX_LAYERS);
  PG_REQUIRE(d->Hq >= 32 && d->Hq % 32 == 0 && d->Hd >= 32 && d->Hd % 32 == 0 && d->Dz >= 8 && d->Dz % 8 == 0,
             "hidden sizes must be multiples of 32 and d_z a multiple of 8 (got Hq=%d Hd=%d Dz=%d)", d->Hq, d->Hd, d->Dz);
  PG_REQUIRE(d->Lq >= 0 && d->Lq <= PG_MAX_ENC_LAYERS && (d->qbidir == 0 || d->qbidir == 1),
             "encoder shape: Lq=%d outside [0,%d] or qbidir=%d not 0/1", d->Lq, PG_MAX_ENC_LAYERS, d->qbidir);
  PG_REQUIRE(d->pad >= 0 && d->pad < d->V && d->bos >= 0 && d->bos < d->V && d->eos >= 0 && d->eos < d->V,
             "special token ids must lie in [0,V)");
  return PG_OK;
}

static inline ParamOff param_offsets(const pg_dims* d) {
  ParamOff o;
  long p = 0;
  auto take = [&](long n) { long r = p; p += n; return r; };
  o.x_emb = take((long)d->V * d->E);
  for (int l = 0; l < PG_MAX_ENC_LAYERS; ++l)
    for (int r = 0; r < 2; ++r) o.g_wih[l][r] = o.g_whh[l][r] = o.g_bih[l][r] = o.g_bhh[l][r] = -1;
  for (int l = 0; l < enc_layers(d); ++l)
    for (int r = 0; r < enc_dirs(d); ++r) {
      o.g_wih[l][r] = take(3L * d->Hq * enc_in(d, l));
      o.g_whh[l][r] = take(3L * d->Hq * d->Hq);
      o.g_bih[l][r] = take(3L * d->Hq);
      o.g_bhh[l][r] = take(3L * d->Hq);
    }
  o.enc_wih = o.g_wih[0][0]; o.enc_whh = o.g_whh[0][0]; o.enc_bih = o.g_bih[0][0]; o.enc_bhh = o.g_bhh[0][0];
  const long q_in = (long)d->Hq * enc_dirs(d);
  o.qmu_w = take((long)d->Dz * q_in);
  o.qmu_b = take(d->Dz);
  o.qlv_w = take((long)d->Dz * q_in);
  o.qlv_b = take(d->Dz);
  o.lat_w = take((long)d->Hd * d->Dz);
  o.lat_b = take(d->Hd);
  for (int l = 0; l < d->L; ++l) {
    long in = l == 0 ? d->E + d->Dz : d->Hd;
    o.dec_wih[l] = take(3L * d->Hd * in);
    o.dec_whh[l] = take(3L * d->Hd * d->Hd);
    o.dec_bih[l] = take(3L * d->Hd);
    o.dec_bhh[l] = take(3L * d->Hd);
  }
  o.fc_w = take((long)d->V * d->Hd);
  o.fc_b = take(d->V);
  o.total = p;
  return o;
}

// byte offsets into the shadow buffer
struct ShadowOff {
  size_t enc_whh, enc_whhT, wq, wqT, lat_w, lat_wT, wih0z, wih0zT;
  size_t whh[PG_MAX_LAYERS], whhT[PG_MAX_LAYERS], wih[PG_MAX_LAYERS], wihT[PG_MAX_LAYERS];
  size_t fc_w, fc_wT;
  size_t tab_enc, tab_dec, bias_gi[PG_MAX_LAYERS], bq, fc_b;
  size_t total;
};

static inline ShadowOff shadow_offsets(const pg_dims* d, int dtype) {
  const size_t es = dtype == PG_BF16 ? 2 : 4;
  ShadowOff s;
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t r = p; p = align_up(p + bytes, 256); return r; };
  const size_t Hq = d->Hq, Hd = d->Hd, Dz = d->Dz, Vp = vpad(d);
  s.enc_whh = take(3 * Hq * Hq * es); s.enc_whhT = take(3 * Hq * Hq * es);
  s.wq = take(2 * Dz * Hq * es); s.wqT = take(2 * Dz * Hq * es);
  s.lat_w = take(Hd * Dz * es); s.lat_wT = take(Hd * Dz * es);
  s.wih0z = take(3 * Hd * Dz * es); s.wih0zT = take(3 * Hd * Dz * es);
  for (int l = 0; l < d->L; ++l) {
    s.whh[l] = take(3 * Hd * Hd * es); s.whhT[l] = take(3 * Hd * Hd * es);
    if (l > 0) { s.wih[l] = take(3 * Hd * Hd * es); s.wihT[l] = take(3 * Hd * Hd * es); }
    else { s.wih[l] = s.wihT[l] = 0; }
  }
  s.fc_w = take(Vp * Hd * es); s.fc_wT = take(Vp * Hd * es);
  s.tab_enc = take((size_t)d->V * 3 * Hq * 4);
  s.tab_dec = take((size_t)d->V * 3 * Hd * 4);
  for (int l = 0; l < d->L; ++l) s.bias_gi[l] = take(3 * Hd * 4);
  s.bq = take(2 * Dz * 4);
  s.fc_b = take(Vp * 4);
  s.total = p;
  return s;
}

// Packed (time-major, length-sorted) schedule of one batch: rows of step t are the prefix b < B[t] and sit at
// packed rows off[t] .. off[t]+B[t]-1 (the layout of torch's PackedSequence).
struct Sched {
  int T;                 // number of steps
  int off[PG_MAX_T + 2]; // off[T] = total rows
};
static inline int sched_B(const Sched& s, int t) { return s.off[t + 1] - s.off[t]; }

static inline int make_sched(const int32_t* lens, int B, int Tmax, int drop_last, Sched* s) {
  PG_REQUIRE(B >= 1, "empty batch");
  PG_REQUIRE(lens != nullptr, "lengths is NULL");
  for (int b = 0; b < B; ++b) {
    PG_REQUIRE(lens[b] >= 1 && lens[b] <= Tmax, "length[%d]=%d outside [1,Tmax=%d]", b, lens[b], Tmax);
    PG_REQUIRE(b == 0 || lens[b] <= lens[b - 1], "lengths must be sorted in decreasing order (row %d), as pack_sequence requires", b);
  }
  const int T = lens[0] - (drop_last ? 1 : 0);
  PG_REQUIRE(T <= PG_MAX_T, "sequence length %d exceeds the supported maximum %d", lens[0], PG_MAX_T);
  s->T = T;
  int o = 0, b = B;
  for (int t = 0; t < T; ++t) {
    while (b > 0 && lens[b - 1] - (drop_last ? 1 : 0) <= t) --b;
    s->off[t] = o;
    o += b;
  }
  s->off[T] = o;
  s->off[T + 1] = o;
  return PG_OK;
}

// ---- training workspace ------------------------------------------------------------------------
// (SC_CE_TICKET follows the two CE sums and SC_KL follows SC_KL_SUM: each group is zeroed by ONE memset, and the kernels'
// last block - told by the ticket; SC_KL itself serves as the KL ticket until that block overwrites it - finalises the loss)
enum { SC_CE_SUM = 0, SC_CE_CNT, SC_CE_TICKET, SC_KL_SUM, SC_KL, SC_RECON, SC_INV_CNT, SC_GS_CE, SC_GS_KL, SC_N = 16 };

struct TrainWs {
  // int32 arrays
  size_t enc_tok, dec_tok, dec_tgt, dec_t, dec_b, dec_off;
  size_t scalars;
  size_t sync, sync_bwd;   // counters of the group-persistent decoder kernels (pg_persist.h), 8256 bytes each; `scalars` follows
  // encoder
  size_t e_carry, e_hop[2], e_sv[5], e_hfin_op, e_mulv, e_eps, e_zop;
  // decoder forward
  size_t d_zop, d_zproj, d_h0_32, d_h0_op, d_carry[PG_MAX_LAYERS], d_hseq[PG_MAX_LAYERS], d_hdrop[PG_MAX_LAYERS];
  size_t d_sv[PG_MAX_LAYERS][5], d_gi, d_logits, d_dlogits;
  // backward
  size_t b_dGI_l[PG_MAX_LAYERS], b_dGH_l[PG_MAX_LAYERS], b_dhz_l[PG_MAX_LAYERS][2];
  size_t b_dout, b_dGI, b_dGH, b_dhz[2], b_dh0, b_dh0_op, b_dzproj, b_dzproj_op, b_dT, e_dT, b_dq, b_dq_op, b_dhfin;
  size_t e_slab;             // the encoder table gradient's partial sums, one row of 3 Hq floats per chunk
  size_t d_ts[3], e_ts[3];   // rows grouped by token for the embedding-table gradients (pg_scatter.h TokSort: perm, chunks, tokchunk)
  size_t b_slab[2];   // split-K slabs of the large weight-gradient GEMMs, one per side stream (PG_TN_SLAB_BYTES, pg_gemm.h)
  size_t total;
};

static inline TrainWs train_ws_layout(const pg_dims* d, int dtype, long B, long Ne, long Nd) {
  const size_t es = dtype == PG_BF16 ? 2 : 4;
  TrainWs w;
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t r = p; p = align_up(p + bytes, 256); return r; };
  const size_t Hq = d->Hq, Hd = d->Hd, Dz = d->Dz, Vp = vpad(d);
  const size_t Hm = Hq > Hd ? Hq : Hd;
  const size_t Nm = Ne > Nd ? Ne : Nd;
  w.enc_tok = take(Ne * 4); w.dec_tok = take(Nd * 4); w.dec_tgt = take(Nd * 4); w.dec_t = take(Nd * 4);
  w.dec_b = take(Nd * 4); w.dec_off = take((PG_MAX_T + 2) * 4);
  // the counters of the decoder's two persistent kernels, then the scalars (whose first three words are the CE sums and
  // their ticket): everything the decoder forward must find zeroed lies in ONE span, cleared by one memset
  w.sync = take(2064 * 4);   // PG_PERSIST_SYNC_WORDS (pg_persist.h): forward recurrence
  w.sync_bwd = take(2064 * 4);   // BPTT
  w.scalars = take(SC_N * 4);
  w.e_carry = take(B * Hq * 4); w.e_hop[0] = take(B * Hq * es); w.e_hop[1] = take(B * Hq * es);
  // r, z, n, hn of the encoder are kept in the persistent kernels' lane-major order, one 16-row slot per (step, block),
  // steps padded by 16 rows (pg_persist.hip): up to 16 * (max steps 512 + 1) rows more than the packed count
  for (int i = 0; i < 4; ++i) w.e_sv[i] = take((Ne + 16 * 513) * Hq * es);
  w.e_sv[4] = take(Ne * Hq * es);
  w.e_hfin_op = take(B * Hq * es); w.e_mulv = take(B * 2 * Dz * 4); w.e_eps = take(B * Dz * 4); w.e_zop = take(B * Dz * es);
  w.d_zop = take(B * Dz * es); w.d_zproj = take(B * 3 * Hd * 4); w.d_h0_32 = take(B * Hd * 4); w.d_h0_op = take(B * Hd * es);
  for (int l = 0; l < d->L; ++l) {
    w.d_carry[l] = take(B * Hd * 4);
    w.d_hseq[l] = take(Nd * Hd * es);
    w.d_hdrop[l] = take(Nd * Hd * es);
    for (int i = 0; i < 4; ++i) w.d_sv[l][i] = take((Nd + 16 * 11 * 512 + 16) * Hd * es);   // lane-order slots (PG_SV_LANE_PAD_ROWS, pg_persist.h)
    w.d_sv[l][4] = take(Nd * Hd * es);
  }
  w.d_gi = take(Nd * 3 * Hd * 4);
  w.d_logits = take(Nd * Vp * 4); w.d_dlogits = take(Nd * Vp * es);
  w.b_dout = take(Nd * Hd * 4);
  w.b_dGI = take(Nm * 3 * Hm * es); w.b_dGH = take(Nm * 3 * Hm * es);
  w.b_dhz[0] = take(B * Hm * 4); w.b_dhz[1] = take(B * Hm * 4);
  for (int l = 0; l < d->L; ++l) {   // the decoder BPTT runs as a wavefront over layers: every layer keeps its own buffers
    w.b_dGI_l[l] = take(Nd * 3 * Hd * es); w.b_dGH_l[l] = take(Nd * 3 * Hd * es);
    w.b_dhz_l[l][0] = take(B * Hd * 4); w.b_dhz_l[l][1] = take(B * Hd * 4);
  }
  w.b_dh0 = take(B * Hd * 4); w.b_dh0_op = take(B * Hd * es);
  w.b_dzproj = take(B * 3 * Hd * 4); w.b_dzproj_op = take(B * 3 * Hd * es);
  w.b_dT = take(Vp * 3 * Hd * 4);
  // the encoder's own copies: its embedding-gradient tail runs beside the decoder's deferred one (pg_vae.hip)
  w.e_dT = take(Vp * 3 * Hq * 4);
  // (chunk tables: N / 128 + Vp entries of 16 bytes - pg_scatter.h tok_sort_chunks, PG_TS_ROWS = 128, checked in pg_vae.hip)
  w.d_ts[0] = take(Nd * 4); w.d_ts[1] = take((Nd / 128 + Vp) * 16); w.d_ts[2] = take((Vp + 1) * 4);
  w.e_ts[0] = take(Ne * 4); w.e_ts[1] = take((Ne / 128 + Vp) * 16); w.e_ts[2] = take((Vp + 1) * 4);
  w.e_slab = take((Ne / 128 + Vp) * 3 * Hq * 4);
  w.b_dq = take(B * 2 * Dz * 4); w.b_dq_op = take(B * 2 * Dz * es); w.b_dhfin = take(B * Hq * 4);
  w.b_slab[0] = take((size_t)776 * 128 * 128 * 4); w.b_slab[1] = take((size_t)776 * 128 * 128 * 4);
  w.total = p;
  return w;
}

struct SampleWs {
  size_t zop, zproj, h0_32, h0_op, carry[PG_MAX_LAYERS], hop[PG_MAX_LAYERS][2], tok, eos, sync;
  size_t total;
};
static inline SampleWs sample_ws_layout(const pg_dims* d, int dtype, long n) {
  const size_t es = dtype == PG_BF16 ? 2 : 4;
  SampleWs w;
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t r = p; p = align_up(p + bytes, 256); return r; };
  const size_t Hd = d->Hd, Dz = d->Dz;
  w.zop = take(n * Dz * es); w.zproj = take(n * 3 * Hd * 4); w.h0_32 = take(n * Hd * 4); w.h0_op = take(n * Hd * es);
  for (int l = 0; l < d->L; ++l) {
    w.carry[l] = take(n * Hd * 4);
    w.hop[l][0] = take(n * Hd * es); w.hop[l][1] = take(n * Hd * es);
  }
  w.tok = take(n * 4); w.eos = take(n * 4);
  w.sync = take(4112 * 4);   // PG_SAMPLE_SYNC_WORDS (pg_persist.h)
  w.total = p;
  return w;
}
