"""GPU parity with a vocabulary above 64 symbols (PG_MAX_V = 128, PG_VPAD = 128: plain-GEMM projection, two vocabulary
entries per lane in the cross-entropy and draw kernels, per-step sampler) against a fixture generated from the REAL
reference with OneHotVocab.from_data over 92 characters (tests/golden/train_wide.npz, oracle/make_golden.py) and against
the oracle at the reference's hidden sizes."""
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as O
from tests.helpers import REAL, TINY, canon_named, rel_err

pytestmark = pytest.mark.gpu


def _model(g, dims, seed, dt, **kw):
    from polygon_amd.vae.vae_model import VAE, get_vocabulary
    alphabet = [str(c) for c in g["alphabet"]]
    vocab = get_vocabulary(alphabet[:-4])            # from_data over the same characters
    assert len(vocab) == 96
    P = O.init_params(seed=seed, V=96, **dims)
    model = VAE(vocabulary=vocab, q_d_h=dims["Hq"], d_z=dims["Dz"], d_d_h=dims["Hd"], d_n_layers=dims["L"], compute_dtype=dt, **kw)
    named = canon_named(model)
    assert set(named) == set(P)
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(torch.from_numpy(P[k]))
    return model.cuda(), {k: v.astype(np.float64) for k, v in P.items()}, alphabet


@pytest.fixture(scope="module")
def wide(golden_dir):
    return np.load(os.path.join(golden_dir, "train_wide.npz"))


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_forward_gradients_and_tokeniser_match_reference(wide, dt):
    g = wide
    model, _, _ = _model(g, TINY, 21, dt)
    assert [model.bos, model.eos, model.pad, model.unk] == g["specials"].tolist()
    model.train()
    batch = model.get_collate_fn()([str(s) for s in g["strings"]])
    assert batch.lengths.tolist() == g["lens"].tolist()
    assert (batch.tokens.cpu().numpy() == g["ids"]).all()
    eps = torch.from_numpy(g["eps"][0]).cuda()
    z, kl, mu = model.forward_encoder(batch, return_mu=True, eps=eps)
    recon, y = model.forward_decoder(batch, z, return_y=True)
    tol = 1e-3 if dt == "fp32" else 2e-2
    assert y.shape == g["y"].shape and y.shape[2] == 96
    assert rel_err(z, g["z"]) <= tol and rel_err(y, g["y"]) <= tol
    assert abs(kl.item() - float(g["kl"])) <= tol * abs(float(g["kl"]))
    assert abs(recon.item() - float(g["recon"])) <= tol * abs(float(g["recon"]))
    model._eps_override = eps
    kl, recon = model(batch)
    (float(g["kl_w"]) * kl + recon).backward()
    for k, p in canon_named(model).items():
        assert rel_err(p.grad, g["grad/" + k]) <= 1e-2, (k, rel_err(p.grad, g["grad/" + k]))


def test_fused_trajectory_matches_reference(wide):
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    g = wide
    model, _, _ = _model(g, TINY, 21, "fp32")
    model.train()
    batch = model.get_collate_fn()([str(s) for s in g["strings"]])
    opt = FusedClipAdam(model, lr=3e-4, clip_grad=50)
    for s in range(3):
        kl, recon = model._engine.train_step(batch, float(g["kl_w"]), opt.flat_grad, eps=torch.from_numpy(g["eps"][s]).cuda())
        opt.step()
        assert abs(kl.item() - g["losses"][s, 0]) <= 1e-3 * abs(g["losses"][s, 0])
        assert abs(recon.item() - g["losses"][s, 1]) <= 1e-3 * abs(g["losses"][s, 1])
        assert abs(opt.gnorm.item() - g["gnorms"][s]) <= 1e-2 * g["gnorms"][s]
    for k, p in canon_named(model).items():
        assert rel_err(p, g["pfinal/" + k]) <= 1e-3, k


def test_sampler_strings_identical_to_reference(wide):
    """fp32: the pinned inverse-CDF draw over 96 entries gives the reference's strings for the fixture's uniforms, and
    argmax decoding gives its argmax strings; the probability dump is 128 wide with zeros above V."""
    g = wide
    model, P, alphabet = _model(g, TINY, 21, "fp32")
    model.eval()
    n, ml = g["sample_z"].shape[0], int(g["sample_max_len"])
    z = torch.from_numpy(g["sample_z"])
    t0, e0 = model.sample_tokens(n, max_len=ml, z=z, uniforms=torch.from_numpy(g["sample_u"]))
    got = model.vocabulary.decode_batch(t0.cpu().numpy(), e0.cpu().numpy())
    assert got == [str(s) for s in g["sample_strings"]]
    assert model.sample(n, max_len=ml, z=z, multinomial=False) == [str(s) for s in g["sample_strings_argmax"]]
    tok, ep, probs = model.sample_tokens(n, max_len=ml, z=z, uniforms=torch.from_numpy(g["sample_u"]), return_probs=True)
    assert tuple(probs.shape) == (ml - 1, n, 128)
    with O.vocabulary(alphabet):
        x, _, plog = O.sample_tokens(P, g["sample_z"].astype(np.float64), g["sample_u"], max_len=ml, return_probs=True)
    assert (tok.cpu().numpy() == x).all()
    pr = probs.cpu().numpy()
    assert (pr[:, :, 96:] == 0).all()
    live = np.array([[i + 1 < ep[r].item() for r in range(n)] for i in range(ml - 1)])      # steps a row really decoded
    assert np.abs(pr[:, :, :96] - np.stack(plog))[live].max() <= 1e-5


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_reference_sized_model_on_a_ragged_batch_matches_oracle(wide, dt):
    """Hd 512 / Hq 256: in bf16 the persistent recurrences run (96-row gather tables), the projection and cross entropy
    take the plain-GEMM + two-entries-per-lane path; d_dropout on."""
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    from tests.helpers import dropout_masks
    g = wide
    model, P, alphabet = _model(g, REAL, 33, dt, d_dropout=0.2)
    model.train()
    rng = np.random.default_rng(8)
    chars = alphabet[:-4]
    strings = ["".join(rng.choice(chars, size=int(k))) for k in rng.integers(1, 60, size=45)] + ["", chars[0], chars[-1] * 90]
    n = len(strings)
    batch = model.get_collate_fn()(list(strings))
    eps = rng.standard_normal((n, REAL["Dz"])).astype(np.float32)
    opt = FusedClipAdam(model, lr=0.0)
    kl, rec = model._engine.train_step(batch, 0.5, opt.flat_grad, eps=torch.from_numpy(eps).cuda(), p_drop=0.2, seed=31337)
    with O.vocabulary(alphabet):
        seqs = O.collate(strings)
        T = max(len(s) for s in seqs)
        dm = dropout_masks(31337, 0.2, REAL["L"] - 1, n, T, REAL["Hd"])
        enc, dec = O.vae_forward(P, seqs, eps.astype(np.float64), dm)
        G, _ = O.vae_backward(P, enc, dec, g_kl=0.5, g_recon=1.0)
    tol = 1e-3 if dt == "fp32" else 2e-2
    assert abs(kl.item() - enc["kl"]) <= tol * abs(enc["kl"]) and abs(rec.item() - dec["recon"]) <= tol * abs(dec["recon"])
    worst = max((rel_err(gv, G[name]), name) for name, gv in zip(O.param_order(P), model._engine.grad_views(opt.flat_grad)))
    assert worst[0] <= 1e-2, worst
    # bf16 sampler on the wide vocabulary: the per-step kernels (the one-launch sampler covers V <= 64); tokens in range
    model.eval()
    tok, ep = model.sample_tokens(64, max_len=40, seed=5)
    assert int(tok.max()) < 96 and int(ep.min()) >= 2


def test_vocabulary_above_the_limit_is_rejected_by_name():
    from polygon_amd import _lib
    from polygon_amd.vae.vae_model import VAE, get_vocabulary
    vocab = get_vocabulary([chr(c) for c in range(0x4E00, 0x4E00 + 130)])      # 130 characters + 4 specials
    model = VAE(vocabulary=vocab, q_d_h=32, d_z=16, d_d_h=64).cuda()
    with pytest.raises(_lib.PolygonHipError, match="vocabulary size 134 outside"):
        model.sample(2, max_len=5)
