"""Oracle pins of the kernels the benchmark measures.

* configs[1] trains with d_dropout 0.2.  torch's nn.GRU dropout stream cannot be reproduced from outside, so the masks the
  HIP kernels apply (pg_dropout_masks_host: the kernels' own counter hash, evaluated on the host) are handed to the CPU
  oracle as ``drop_masks``; losses and all 25 gradients then have to agree to north_star's tolerances, in fp32 and bf16,
  on the per-step and on the persistent kernels.
* The bf16 samplers (per-step and one-launch) are run TEACHER-FORCED on the token stream the fp64 oracle draws from the
  golden z / uniforms (pg_sample_forced), which makes every step comparable: probabilities within 2e-2, and the drawn
  token identical wherever the uniform is further than that from a CDF edge.
"""
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as O
from polygon_amd import _lib
from tests.helpers import REAL, build_model, dropout_masks, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt,mask", [("fp32", 0), ("bf16", 0), ("bf16", 7)])
def test_dropout_path_matches_oracle(dt, mask):
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    p_drop, seed, n = 0.2, 20250310, 48
    model, P = build_model(REAL, 17, dt, d_dropout=p_drop)
    model.train()
    strings = list(O.synthetic_smiles(n - 3, seed=9)) + ["", "C", "C" * 97]
    batch = model.get_collate_fn()(list(strings))
    eps = np.random.default_rng(5).standard_normal((n, REAL["Dz"])).astype(np.float32)
    opt = FusedClipAdam(model, lr=0.0)
    L = _lib.lib()
    old = L.pg_persist_disable(mask)
    try:
        kl, rec = model._engine.train_step(batch, 0.5, opt.flat_grad, eps=torch.from_numpy(eps).cuda(), p_drop=p_drop, seed=seed)
        torch.cuda.synchronize()
    finally:
        L.pg_persist_disable(old)
    seqs = O.collate(strings)
    T = max(len(s) for s in seqs)
    dm = dropout_masks(seed, p_drop, REAL["L"] - 1, n, T, REAL["Hd"])
    frac = float(np.mean([(m == 0).mean() for m in dm]))
    assert abs(frac - p_drop) < 5e-3, frac                         # the hash drops p of the entries ...
    assert set(np.unique(dm[0])) == {0.0, np.float32(1.0 / (1.0 - np.float32(p_drop)))}   # ... and scales the rest by 1/(1-p)
    enc, dec = O.vae_forward(P, seqs, eps.astype(np.float64), dm)
    G, _ = O.vae_backward(P, enc, dec, g_kl=0.5, g_recon=1.0)
    # the masks matter: without them the oracle's decoder gradients are 5-15 % off, far outside the tolerance (guards
    # against a vacuous test; the LOSS of a randomly initialised model barely moves under dropout)
    enc0, dec0 = O.vae_forward(P, seqs, eps.astype(np.float64), None)
    G0, _ = O.vae_backward(P, enc0, dec0, g_kl=0.5, g_recon=1.0)
    for name in ("decoder_lat.weight", "decoder_rnn.weight_ih_l1", "decoder_rnn.weight_hh_l0", "decoder_fc.weight"):
        assert rel_err(G0[name], G[name]) > 4e-2, name
    tol = 1e-3 if dt == "fp32" else 2e-2
    assert abs(kl.item() - enc["kl"]) <= tol * abs(enc["kl"])
    assert abs(rec.item() - dec["recon"]) <= (1e-3 if dt == "fp32" else 1e-3) * abs(dec["recon"])   # bf16 measured 2e-6
    for name, g in zip(O.param_order(P), model._engine.grad_views(opt.flat_grad)):
        assert rel_err(g, G[name]) <= 1e-2, (name, rel_err(g, G[name]))          # measured: fp32 5e-7, bf16 <= 4e-3


@pytest.mark.parametrize("golden_bias", [True, False], ids=["golden_eos_bias", "long_rows"])
@pytest.mark.parametrize("mask", [8, 0], ids=["per_step", "one_launch"])
def test_bf16_sampler_teacher_forced_vs_oracle(golden_dir, mask, golden_bias):
    g = np.load(os.path.join(golden_dir, "sample_real.npz"))
    # golden_bias: the eos-biased head the golden strings were drawn with (short rows); otherwise the plain random head,
    # whose rows run ~50 steps, so that many more recorded draws are compared
    model, P = build_model(REAL, 4, "bf16", eos_bias=float(g["eos_bias"]) if golden_bias else 0.0)
    model.eval()
    max_len, temp = int(g["max_len"]), float(g["temp"])
    # rows 0..15: the golden z / uniforms of the real reference; 48 more seeded rows (rows are independent) -> n = 64
    rng = np.random.default_rng(64)
    n0, n = g["z"].shape[0], 64
    z = np.concatenate([g["z"], rng.standard_normal((n - n0, REAL["Dz"])).astype(np.float32)])
    u = np.concatenate([g["u"], rng.random((max_len - 1, n - n0), dtype=np.float32)], axis=1)
    # the oracle reproduces the reference's golden strings from these inputs (also tests/test_oracle_golden.py); here it
    # hands out the probabilities of every step and the tokens it fed as well
    x, ep, probs, fed = O.sample_tokens(P, z.astype(np.float64), u, max_len=max_len, temp=temp, return_probs=True, return_fed=True)
    if golden_bias:
        assert O.tokens_to_strings(x, ep)[:n0] == [str(s) for s in g["strings"]]
    probs = np.stack(probs)                                        # [T, n, V]
    V = probs.shape[2]
    L = _lib.lib()
    old = L.pg_persist_disable(mask)
    try:
        tok, epg, pg = model.sample_tokens(n, max_len=max_len, z=torch.from_numpy(z), temp=temp, uniforms=torch.from_numpy(u),
                                           forced=torch.from_numpy(fed.astype(np.uint8)), return_probs=True)
        torch.cuda.synchronize()
        free_tok, free_ep = model.sample_tokens(n, max_len=max_len, z=torch.from_numpy(z), temp=temp, uniforms=torch.from_numpy(u))
    finally:
        L.pg_persist_disable(old)
    pg = pg.cpu().numpy().astype(np.float64)
    assert (pg[:, :, V:] == 0).all()
    err = np.abs(pg[:, :, :V] - probs).max()
    assert err <= 5e-3, err                                        # north_star allows 2e-2 in bf16; measured 3.4e-4 / 6.3e-5
    assert np.abs(pg[:, :, :V].sum(2) - 1.0).max() <= 1e-5
    # drawn tokens: the draw is "first k with cdf_k > u", so wherever u is further from EVERY oracle CDF edge than the
    # GPU's CDF is from the oracle's at that step, both sides must draw the same token ("identical streams for identical
    # uniforms" up to bf16 rounding of the probabilities themselves)
    cdf = np.cumsum(probs.astype(np.float32), axis=2, dtype=np.float32).astype(np.float64)
    cdf_g = np.cumsum(pg[:, :, :V].astype(np.float32), axis=2, dtype=np.float32).astype(np.float64)
    margin = np.abs(cdf - u[:, :, None].astype(np.float64)).min(2)           # [T, n]
    slack = np.abs(cdf_g - cdf).max(2) + 1e-6
    tokg, epg = tok.cpu().numpy().astype(np.int64), epg.cpu().numpy()
    cols = np.arange(max_len)[None, :]
    live = cols < ep[:, None]                                      # recorded positions (up to and including eos)
    safe = np.ones((n, max_len), dtype=bool)
    safe[:, 1:] = (margin > slack).T
    chk = live & safe & (cols >= 1)
    assert chk.sum() > 0.9 * (live & (cols >= 1)).sum()                            # the comparison covers nearly all of the stream
    # a row's recorded stream is comparable up to its first unsafe draw (a different draw there may end the row early)
    first_unsafe = np.where((~safe & live).any(1), (~safe & live).argmax(1), max_len)
    chk &= cols < first_unsafe[:, None]
    assert (tokg[chk] == x[chk]).all(), float((tokg[chk] != x[chk]).mean())
    # free-running decode with the same uniforms: not claimed identical in bf16 - report, and require that
    # most rows still match the reference's strings
    got = model.vocabulary.decode_batch(free_tok.cpu().numpy(), free_ep.cpu().numpy())
    same = np.mean([a == b for a, b in zip(got, O.tokens_to_strings(x, ep))])
    print("bf16 free-running rows identical to the reference: %.3f (max prob err teacher-forced %.2e)" % (same, err))
    assert same >= 0.95                                            # measured 1.000 (eos-biased head) / 0.984 (plain head: 1 of 64 rows)
