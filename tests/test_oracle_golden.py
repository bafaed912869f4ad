"""Pins the CPU oracle (oracle/vae_oracle.py) against fixtures generated from the REAL reference
(oracle/make_golden.py, run in the build container).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import vae_oracle as O

TINY = dict(Hq=32, Dz=16, Hd=64, L=3)
REAL = dict(Hq=256, Dz=128, Hd=512, L=3)


def _P(seed, dims, dtype=np.float64):
    return {k: v.astype(dtype) for k, v in O.init_params(seed=seed, **dims).items()}


@pytest.fixture(scope="module")
def vocab(golden_dir):
    with open(os.path.join(golden_dir, "vocab.json")) as f:
        return json.load(f)


def test_vocab_table(vocab):
    assert O.ALL_SYMS == vocab["all_syms"]
    assert (O.BOS, O.EOS, O.PAD, O.UNK) == (vocab["bos"], vocab["eos"], vocab["pad"], vocab["unk"])
    assert O.N_VOCAB == 51


def test_string2ids(vocab):
    for s, ids in vocab["string2ids"]:
        assert O.string2ids(s, True, True) == ids, s


def test_ids2string(vocab):
    for ids, s in vocab["ids2string"]:
        if isinstance(s, dict):
            with pytest.raises(IndexError):
                O.ids2string(ids)
        else:
            assert O.ids2string(ids) == s, ids


def test_collate_order(vocab):
    got = O.collate(vocab["collate_in"])
    assert [g.tolist() for g in got] == vocab["collate_ids"]


def test_schedulers(vocab):
    n, s, w0, w1 = vocab["kl_annealer"]["args"]
    for i, v in enumerate(vocab["kl_annealer"]["values"]):
        assert O.kl_weight(i, n, s, w0, w1) == pytest.approx(v, rel=1e-12)
    sg = vocab["sgdr"]
    tab = O.sgdr_lr_table(sg["base_lr"], len(sg["lrs"]) - 1, sg["lr_n_period"], sg["lr_n_mult"], sg["lr_end"])
    np.testing.assert_allclose(tab, sg["lrs"], rtol=1e-12)


def test_param_order(vocab):
    P = O.init_params(0)
    # reference names are relative to model.vae (ModuleList indices); compare shapes + count
    names = O.param_order(P)
    assert [list(P[n].shape) for n in names] == vocab["param_shapes"]
    assert sum(P[n].size for n in names) == vocab["n_params"] == 4614236


ENC_SHAPES = {"enc_bi": dict(TINY, Lq=1, bidir=True), "enc_l2": dict(TINY, Lq=2, bidir=False),
              "enc_bi3": dict(TINY, Lq=3, bidir=True)}     # vae_model.py:63-70 with q_n_layers / q_bidir off their defaults


def test_param_order_of_other_encoder_shapes():
    """nn.GRU lists weight_ih, weight_hh, bias_ih, bias_hh per layer, forward then reverse; q_mu / q_logvar read Hq*dirs."""
    P = O.init_params(0, **ENC_SHAPES["enc_bi3"])
    names = O.param_order(P)
    assert names[:9] == ["x_emb.weight", "encoder_rnn.weight_ih_l0", "encoder_rnn.weight_hh_l0", "encoder_rnn.bias_ih_l0",
                         "encoder_rnn.bias_hh_l0", "encoder_rnn.weight_ih_l0_reverse", "encoder_rnn.weight_hh_l0_reverse",
                         "encoder_rnn.bias_ih_l0_reverse", "encoder_rnn.bias_hh_l0_reverse"]
    assert set(names) == set(P) and len(names) == len(P) == 1 + 4 * 3 * 2 + 6 + 4 * 3 + 2
    assert P["encoder_rnn.weight_ih_l1_reverse"].shape == (96, 64) and P["q_mu.weight"].shape == (16, 64)
    # the default shape still draws what the older fixtures were generated with
    assert O.param_order(O.init_params(0, **TINY))[:11] == O.PARAM_ORDER_HEAD


@pytest.mark.parametrize("tag,dims,seed,full", [("tiny", TINY, 1, True), ("real", REAL, 2, False)] +
                         [(t, d, 11, True) for t, d in ENC_SHAPES.items()])
def test_train_step_matches_reference(golden_dir, tag, dims, seed, full):
    g = np.load(os.path.join(golden_dir, "train_%s.npz" % tag))
    P = _P(seed, dims)
    seqs = O.collate([str(s) for s in g["strings"]])
    kl_w = float(g["kl_w"])
    eps = g["eps"].astype(np.float64)
    enc, dec = O.vae_forward(P, seqs, eps[0])
    np.testing.assert_allclose(enc["z"], g["z"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(enc["mu"], g["mu"], rtol=2e-4, atol=2e-5)
    assert enc["kl"] == pytest.approx(float(g["kl"]), rel=1e-5)
    assert dec["recon"] == pytest.approx(float(g["recon"]), rel=1e-5)
    np.testing.assert_allclose(dec["y"], g["y"], rtol=1e-3, atol=1e-4)
    G, _ = O.vae_backward(P, enc, dec, g_kl=kl_w, g_recon=1.0)
    for k in P:
        if full:
            ref = g["grad/" + k]
            np.testing.assert_allclose(G[k], ref, rtol=1e-3, atol=1e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)
        else:
            assert np.linalg.norm(G[k]) == pytest.approx(float(g["gnorm/" + k]), rel=1e-4), k
            sl = G[k].reshape(-1)[:: max(1, G[k].size // 257)][:257]
            ref = g["gslice/" + k]
            np.testing.assert_allclose(sl, ref, rtol=1e-3, atol=1e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)
    # 3 steps of clip + Adam (vae_trainer.py:133-138)
    st = O.adam_state(P)
    for s in range(len(g["losses"])):
        enc, dec = O.vae_forward(P, seqs, eps[s])
        assert enc["kl"] == pytest.approx(g["losses"][s, 0], rel=2e-4)
        assert dec["recon"] == pytest.approx(g["losses"][s, 1], rel=2e-4)
        G, _ = O.vae_backward(P, enc, dec, g_kl=kl_w, g_recon=1.0)
        G, total = O.clip_grad_norm(G, 50.0)
        assert total == pytest.approx(g["gnorms"][s], rel=1e-3)
        P = O.adam_step(P, G, st, lr=3e-4)
    for k in P:
        ref = g["pfinal/" + k]
        got = P[k] if full else P[k].reshape(-1)[:: max(1, P[k].size // 257)][:257]
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-5, err_msg=k)


def test_wide_vocabulary_matches_reference(golden_dir):
    """A data-derived vocabulary of 96 symbols (OneHotVocab.from_data, vae_model.py:9-20; above the 64 one wave lane each
    covers): tokeniser ids, forward, every gradient, the 3-step trajectory and the sampler's strings of the REAL
    reference (oracle/make_golden.py wide_vocab_fixture)."""
    g = np.load(os.path.join(golden_dir, "train_wide.npz"))
    alphabet = [str(c) for c in g["alphabet"]]
    with O.vocabulary(alphabet):
        assert O.ALL_SYMS == alphabet and O.N_VOCAB == 96
        assert [O.BOS, O.EOS, O.PAD, O.UNK] == g["specials"].tolist()
        seqs = O.collate([str(s) for s in g["strings"]])
        assert [len(s) for s in seqs] == g["lens"].tolist()
        for i, s in enumerate(seqs):
            assert s.tolist() == g["ids"][i, :len(s)].tolist()
        P = {k: v.astype(np.float64) for k, v in O.init_params(seed=21, V=96, **TINY).items()}
        eps = g["eps"].astype(np.float64)
        enc, dec = O.vae_forward(P, seqs, eps[0])
        np.testing.assert_allclose(enc["z"], g["z"], rtol=2e-4, atol=2e-5)
        assert enc["kl"] == pytest.approx(float(g["kl"]), rel=1e-5) and dec["recon"] == pytest.approx(float(g["recon"]), rel=1e-5)
        np.testing.assert_allclose(dec["y"], g["y"], rtol=1e-3, atol=1e-4)
        G, _ = O.vae_backward(P, enc, dec, g_kl=float(g["kl_w"]), g_recon=1.0)
        for k in P:
            ref = g["grad/" + k]
            np.testing.assert_allclose(G[k], ref, rtol=1e-3, atol=1e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)
        P0 = {k: v.copy() for k, v in P.items()}
        st = O.adam_state(P)
        for s in range(len(g["losses"])):
            enc, dec = O.vae_forward(P, seqs, eps[s])
            G, _ = O.vae_backward(P, enc, dec, g_kl=float(g["kl_w"]), g_recon=1.0)
            G, total = O.clip_grad_norm(G, 50.0)
            assert total == pytest.approx(g["gnorms"][s], rel=1e-3)
            P = O.adam_step(P, G, st, lr=3e-4)
        for k in P:
            np.testing.assert_allclose(P[k], g["pfinal/" + k], rtol=1e-3, atol=2e-5, err_msg=k)
        ml = int(g["sample_max_len"])
        x, ep = O.sample_tokens(P0, g["sample_z"].astype(np.float64), g["sample_u"], max_len=ml)
        assert O.tokens_to_strings(x, ep) == [str(s) for s in g["sample_strings"]]
        x, ep = O.sample_tokens(P0, g["sample_z"].astype(np.float64), None, max_len=ml, multinomial=False)
        assert O.tokens_to_strings(x, ep) == [str(s) for s in g["sample_strings_argmax"]]
    assert O.N_VOCAB == 51 and O.PAD == 49


@pytest.mark.parametrize("tag,dims,seed", [("tiny", TINY, 3), ("real", REAL, 4), ("real_t07", REAL, 5)])
def test_sampler_matches_reference(golden_dir, tag, dims, seed):
    g = np.load(os.path.join(golden_dir, "sample_%s.npz" % tag))
    P = _P(seed, dims)
    P["decoder_fc.bias"] = P["decoder_fc.bias"].copy()
    P["decoder_fc.bias"][O.EOS] += float(g["eos_bias"])
    max_len = int(g["max_len"])
    x, ep = O.sample_tokens(P, g["z"].astype(np.float64), g["u"], max_len=max_len, temp=float(g["temp"]))
    assert O.tokens_to_strings(x, ep) == [str(s) for s in g["strings"]]
    x, ep = O.sample_tokens(P, g["z"].astype(np.float64), None, max_len=max_len, temp=float(g["temp"]),
                            multinomial=False)
    assert O.tokens_to_strings(x, ep) == [str(s) for s in g["strings_argmax"]]


def test_stock_torch_cpu_baseline_matches_oracle():
    """oracle/torch_cpu.py (the CPU baseline bench.py times: the hot path in stock torch.nn ops, as the reference runs it)
    against the numpy oracle: losses, gradient norm and one clip+Adam step."""
    import torch
    from oracle import torch_cpu as TC
    dims = dict(Hq=32, Dz=16, Hd=64, L=3)
    P = O.init_params(seed=5, **dims)
    strings = O.synthetic_smiles(12, seed=2) + ["", "C"]
    seqs = O.collate(strings)
    eps = np.random.default_rng(0).standard_normal((len(seqs), dims["Dz"])).astype(np.float32)
    m = TC.StockVAE(P)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    kl, rec, gn = m.train_step(TC.to_tensors(seqs), 0.5, opt, eps=torch.from_numpy(eps))
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    enc, dec = O.vae_forward(P64, seqs, eps.astype(np.float64))
    G, _ = O.vae_backward(P64, enc, dec, g_kl=0.5, g_recon=1.0)
    Gc, total = O.clip_grad_norm(G, 50.0)
    assert abs(kl - enc["kl"]) <= 1e-4 * abs(enc["kl"]) and abs(rec - dec["recon"]) <= 1e-5 * abs(dec["recon"])
    assert abs(gn - total) <= 1e-4 * total
    st = O.adam_state(P64)
    P1 = O.adam_step(dict(P64), Gc, st, lr=3e-4)
    for k, p in m.named_parameters():
        assert np.abs(p.detach().numpy() - P1[k]).max() <= 2e-5, k
    x, end = m.sample(torch.zeros(3, dims["Dz"]), max_len=12)
    assert x.shape == (3, 12) and (x[:, 0] == O.BOS).all() and (end <= 12).all()
