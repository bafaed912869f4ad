"""GPU parity of the encoder shapes off the reference's defaults (vae_model.py:63-70: q_n_layers > 1, q_bidir=True) --
pg_encoder_gen_fwd / pg_encoder_gen_bwd through the model API and the fused step -- against fixtures generated from the
REAL reference (tests/golden/train_enc_*.npz, oracle/make_golden.py) and against the CPU oracle on seeded, ragged batches.
Tolerances as everywhere: losses 1e-3 fp32 / 2e-2 bf16, gradients 1e-2."""
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as O
from tests.helpers import REAL, TINY, build_model, canon_named, dropout_masks, rel_err

pytestmark = pytest.mark.gpu
SHAPES = {"enc_bi": dict(TINY, Lq=1, bidir=True), "enc_l2": dict(TINY, Lq=2, bidir=False),
          "enc_bi3": dict(TINY, Lq=3, bidir=True)}
TOL = {"fp32": 1e-3, "bf16": 2e-2}


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", sorted(SHAPES))
def test_forward_and_gradients_match_reference(golden_dir, tag, dt):
    g = np.load(os.path.join(golden_dir, "train_%s.npz" % tag))
    model, _ = build_model(SHAPES[tag], 11, dt, q_dropout=0.0)
    model.train()
    batch = model.get_collate_fn()([str(s) for s in g["strings"]])
    eps = torch.from_numpy(g["eps"][0]).cuda()
    z, kl, mu = model.forward_encoder(batch, return_mu=True, eps=eps)
    tol = TOL[dt]
    assert rel_err(mu, g["mu"]) <= tol and rel_err(z, g["z"]) <= tol
    assert abs(kl.item() - float(g["kl"])) <= tol * abs(float(g["kl"]))
    model._eps_override = eps
    kl, recon = model(batch)
    assert abs(recon.item() - float(g["recon"])) <= tol * abs(float(g["recon"]))
    (float(g["kl_w"]) * kl + recon).backward()
    named = canon_named(model)
    assert len(named) == len([k for k in g.files if k.startswith("grad/")])
    for k, p in named.items():
        assert rel_err(p.grad, g["grad/" + k]) <= 1e-2, (k, rel_err(p.grad, g["grad/" + k]))


@pytest.mark.parametrize("tag", sorted(SHAPES))
def test_fused_trajectory_matches_reference(golden_dir, tag):
    """3 fused steps (pg_encoder_gen_fwd / decoder_fwd / decoder_bwd / pg_encoder_gen_bwd / clip_adam) against the
    reference's zero_grad / backward / clip_grad_norm_ / Adam.step trajectory (vae_trainer.py:133-138)."""
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    g = np.load(os.path.join(golden_dir, "train_%s.npz" % tag))
    model, _ = build_model(SHAPES[tag], 11, "fp32", q_dropout=0.0)
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


def _ragged(n, seed):
    return list(O.synthetic_smiles(n - 3, seed=seed)) + ["", "C", "C" * 97]


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("Lq,bidir", [(2, True), (3, False)])
def test_encoder_dropout_and_ragged_batch_match_oracle(dt, Lq, bidir):
    """Reference-sized encoder (Hq 256) with q_dropout: the masks the kernels apply (pg_dropout_masks_host, H = Hq*dirs)
    go to the oracle; empty, one-character and 97-character strings in the batch."""
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    p, seed, n = 0.3, 777, 40
    dims = dict(REAL, Lq=Lq, bidir=bidir)
    model, P = build_model(dims, 23, dt, q_dropout=p)
    model.train()
    strings = _ragged(n, 4)
    batch = model.get_collate_fn()(list(strings))
    eps = np.random.default_rng(6).standard_normal((n, REAL["Dz"])).astype(np.float32)
    opt = FusedClipAdam(model, lr=0.0)
    kl, rec = model._engine.train_step(batch, 0.7, opt.flat_grad, eps=torch.from_numpy(eps).cuda(), q_drop=p, q_seed=seed)
    seqs = O.collate(strings)
    T = max(len(s) for s in seqs)
    W = REAL["Hq"] * (2 if bidir else 1)
    dm = dropout_masks(seed, p, Lq - 1, n, T, W)
    enc, dec = O.vae_forward(P, seqs, eps.astype(np.float64), None, enc_drop_masks=dm)
    G, _ = O.vae_backward(P, enc, dec, g_kl=0.7, g_recon=1.0)
    enc0, dec0 = O.vae_forward(P, seqs, eps.astype(np.float64), None)
    G0, _ = O.vae_backward(P, enc0, dec0, g_kl=0.7, g_recon=1.0)
    assert rel_err(G0["encoder_rnn.weight_hh_l0"], G["encoder_rnn.weight_hh_l0"]) > 4e-2     # the masks matter
    tol = TOL[dt]
    assert abs(kl.item() - enc["kl"]) <= tol * abs(enc["kl"])
    assert abs(rec.item() - dec["recon"]) <= tol * abs(dec["recon"])
    names = O.param_order(P)
    views = model._engine.grad_views(opt.flat_grad)
    assert len(names) == len(views)
    worst = max((rel_err(g, G[name]), name) for name, g in zip(names, views))
    assert worst[0] <= 1e-2, worst


def test_generic_path_on_the_default_shape_equals_the_default_kernels():
    """pg_encoder_gen_* accept the default shape too: same losses and gradients as pg_encoder_fwd / _bwd (fp32)."""
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    model, _ = build_model(REAL, 3, "fp32")
    model.train()
    batch = model.get_collate_fn()(_ragged(33, 8))
    eps = torch.randn(33, REAL["Dz"], generator=torch.Generator().manual_seed(1)).cuda()
    opt = FusedClipAdam(model, lr=0.0)
    eng = model._engine
    kl0, rec0 = [x.item() for x in eng.train_step(batch, 0.4, opt.flat_grad, eps=eps)]
    g0 = opt.flat_grad.clone()
    eng.__class__ = type("GenericEngine", (eng.__class__,), {"generic_encoder": True})
    kl1, rec1 = [x.item() for x in eng.train_step(batch, 0.4, opt.flat_grad, eps=eps)]
    assert abs(kl1 - kl0) <= 1e-5 * abs(kl0) and abs(rec1 - rec0) <= 1e-6 * abs(rec0)
    for v0, v1, (name, _) in zip(eng.grad_views(g0), eng.grad_views(opt.flat_grad), model.vae.named_parameters()):
        assert rel_err(v1, v0) <= 1e-4, name


def test_shards_of_a_batch_sum_to_the_whole_batch():
    """Data parallelism with a multi-layer bidirectional encoder: rank shards normalise by the global
    counts and key the dropout masks by the global row, so their gradients add up to the single-process ones."""
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    dims = dict(Hq=64, Dz=32, Hd=128, L=3, Lq=2, bidir=True)
    model, _ = build_model(dims, 5, "fp32", q_dropout=0.25, d_dropout=0.1)
    model.train()
    batch = model.get_collate_fn()(_ragged(37, 2))
    eps = torch.randn(37, 32, generator=torch.Generator().manual_seed(3)).cuda()
    opt = FusedClipAdam(model, lr=0.0)
    eng = model._engine
    kw = dict(p_drop=0.1, seed=99, q_drop=0.25, q_seed=1234)
    kl, rec = [x.item() for x in eng.train_step(batch, 0.6, opt.flat_grad, eps=eps, **kw)]
    whole = opt.flat_grad.clone()
    acc, kls, recs = torch.zeros_like(whole), 0.0, 0.0
    for r in range(3):
        sh = batch.shard(r, 3)
        k, c = eng.train_step(sh, 0.6, opt.flat_grad, eps=eps[r::3].contiguous(), **kw)
        acc += opt.flat_grad
        kls += k.item()
        recs += c.item()
    assert abs(kls - kl) <= 1e-5 * abs(kl) and abs(recs - rec) <= 1e-5 * abs(rec)
    assert rel_err(acc, whole) <= 1e-4


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_encode_in_eval_mode_matches_oracle(dt):
    """encode() / encode_smiles (vae_model.py:168-181): eval mode, no dropout, mu of the bidirectional 2-layer encoder."""
    dims = dict(REAL, Lq=2, bidir=True)
    model, P = build_model(dims, 29, dt)
    model.eval()
    strings = _ragged(21, 12)
    mu = model.encode_smiles(list(strings), batch_size=8)            # input order, chunks of 8
    seqs = [np.asarray(O.string2ids(s, True, True)) for s in strings]
    order = sorted(range(len(seqs)), key=lambda i: len(strings[i]), reverse=True)
    enc = O.encoder_forward(P, [seqs[i] for i in order], np.zeros((len(seqs), REAL["Dz"])), keep=False)
    ref = np.empty_like(enc["mu"])
    ref[order] = enc["mu"]
    assert rel_err(mu, ref) <= TOL[dt]


def test_cli_train_with_encoder_flags(tmp_path):
    """`polygon train --q_bidir --q_n_layers 2 --q_dropout 0.25` (run.py:354-365 of the reference): the CLI builds the model,
    the fused trainer runs the generic encoder with its dropout, the loss goes down and the checkpoint reloads into a
    model of the same shape (nn.GRU's key list) and encodes."""
    from polygon_amd import run
    from polygon_amd.utils.utils import load_model
    from polygon_amd.vae.vae_model import VAE
    data = tmp_path / "train.smiles"
    data.write_text("\n".join(O.synthetic_smiles(256, seed=18)) + "\n")
    ckpt = tmp_path / "enc.pt"
    run.main(["train", "--train_data", str(data), "--model_save", str(ckpt), "--n_epoch", "3", "--n_batch", "64",
              "--device", "cuda:0", "--compute_dtype", "bf16", "--log_file", str(tmp_path / "log.csv"), "--q_bidir",
              "--q_n_layers", "2", "--q_dropout", "0.25", "--q_d_h", "128", "--kl_start", "0", "--kl_w_start", "0.0",
              "--kl_w_end", "0.05", "--quiet"])
    rows = [l.strip().split(",") for l in open(tmp_path / "log.csv") if l.strip().endswith("Train")]
    assert len(rows) == 3 and float(rows[2][4]) < float(rows[0][4])
    model = load_model(VAE, str(ckpt), "cuda:0", model_params=dict(compute_dtype="bf16", q_bidir=True, q_n_layers=2, q_d_h=128))
    keys = list(model.state_dict().keys())
    assert "encoder_rnn.weight_hh_l1_reverse" in keys and model.q_mu.in_features == 256
    mu = model.encode_smiles(["CCO", "c1ccccc1", "C"])
    assert tuple(mu.shape) == (3, 128) and torch.isfinite(mu).all()
