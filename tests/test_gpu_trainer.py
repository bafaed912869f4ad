"""GPU tests of the fused training step (no autograd) and the trainer shell."""
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as O
from tests.helpers import REAL, TINY, build_model, canon_named, rel_err

pytestmark = pytest.mark.gpu


def test_fused_adam_trajectory_matches_reference(golden_dir):
    """3 fused steps (pg_encoder_fwd/decoder_fwd/decoder_bwd/encoder_bwd/clip_adam) against the reference's trajectory."""
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    g = np.load(os.path.join(golden_dir, "train_tiny.npz"))
    model, _ = build_model(TINY, 1, "fp32")
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


def test_fused_real_dims_trajectory(golden_dir):
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    g = np.load(os.path.join(golden_dir, "train_real.npz"))
    model, _ = build_model(REAL, 2, "fp32")
    model.train()
    batch = model.get_collate_fn()([str(s) for s in g["strings"]])
    opt = FusedClipAdam(model, lr=3e-4, clip_grad=50)
    for s in range(3):
        kl, recon = model._engine.train_step(batch, float(g["kl_w"]), opt.flat_grad, eps=torch.from_numpy(g["eps"][s]).cuda())
        opt.step()
        assert abs(recon.item() - g["losses"][s, 1]) <= 1e-3 * abs(g["losses"][s, 1])
        assert abs(opt.gnorm.item() - g["gnorms"][s]) <= 1e-2 * g["gnorms"][s]
    for k, p in canon_named(model).items():
        a = p.detach().cpu().numpy()
        sl = a.reshape(-1)[:: max(1, a.size // 257)][:257]
        assert rel_err(sl, g["pfinal/" + k]) <= 1e-3, k


def test_dropout_mask_consistent_between_forward_and_backward():
    """Train-mode dropout: the gradient must be that of the SAME masked network.  Checked by finite differences of the
    fused loss along a random direction (the mask is a counter hash, a pure function of (seed, layer, t, row, col))."""
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    model, _ = build_model(TINY, 5, "fp32", d_dropout=0.3)
    model.train()
    batch = model.get_collate_fn()(O.synthetic_smiles(16, seed=9))
    eng = model._engine
    opt = FusedClipAdam(model, lr=0.0)
    eps = torch.from_numpy(np.random.default_rng(4).standard_normal((16, TINY["Dz"])).astype(np.float32)).cuda()

    def loss_at():
        kl, rec = eng.train_step(batch, 0.3, opt.flat_grad, eps=eps, p_drop=0.3, seed=77)
        return (0.3 * kl + rec).item()
    l0 = loss_at()
    g = opt.flat_grad.clone()
    for _ in range(5):      # the forward is atomics-free apart from the loss sums: repeated calls must agree closely
        assert abs(loss_at() - l0) <= 1e-5 * abs(l0)
        assert (opt.flat_grad - g).abs().max().item() <= 1e-4 * g.abs().max().item()
    d = g / g.norm()          # steepest direction: directional derivative = |g|, well above fp32 noise
    flat = eng.ensure_flat()
    h = 2e-3
    with torch.no_grad():
        flat.add_(d, alpha=h); eng.invalidate(); lp = loss_at()
        flat.add_(d, alpha=-2 * h); eng.invalidate(); lm = loss_at()
        flat.add_(d, alpha=h); eng.invalidate()
    fd = (lp - lm) / (2 * h)
    an = float((g * d).sum())
    assert abs(fd - an) <= 2e-2 * abs(an), (fd, an)
    # and dropout actually drops: loss differs from the p=0 loss
    kl, rec = eng.train_step(batch, 0.3, opt.flat_grad, eps=eps, p_drop=0.0, seed=77)
    assert abs((0.3 * kl + rec).item() - loss_at()) > 1e-4


def test_trainer_fit_runs_and_learns(tmp_path):
    from polygon_amd.vae.vae_trainer import VAETrainer
    torch.manual_seed(0)
    model, _ = build_model(TINY, 3, "fp32")
    data = O.synthetic_smiles(96, seed=2)
    log = tmp_path / "log.csv"
    tr = VAETrainer(model, n_batch=32, n_epoch=4, kl_start=0, kl_w_start=0.0, kl_w_end=0.1, lr_start=3e-3,
                    log_file=str(log), model_save=str(tmp_path / "m.pt"), save_frequency=2)
    tr.fit(list(data), list(data[:32]))
    rows = [l.strip().split(",") for l in open(log) if l.strip().endswith(("Train", "Eval"))]
    train = [float(r[4]) for r in rows if r[-1] == "Train"]
    assert len(train) == 4 and train[-1] < train[0]
    assert os.path.exists(tmp_path / "m.pt") and os.path.exists(tmp_path / "m_000.pt") and os.path.exists(tmp_path / "m_002.pt")
    from polygon_amd.utils.utils import load_model
    from polygon_amd.vae.vae_model import VAE
    m2 = load_model(VAE, str(tmp_path / "m.pt"), "cuda", model_params=dict(q_d_h=32, d_z=16, d_d_h=64, d_n_layers=3))
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k


def test_resume_from_train_state(tmp_path):
    """optimiser moments / step count / schedule position are checkpointed next to the weights and a resumed
    fit continues where the first one stopped (same Adam state, next epoch index, same learning rate)."""
    from polygon_amd.utils.utils import load_model
    from polygon_amd.vae.vae_model import VAE
    from polygon_amd.vae.vae_trainer import FusedClipAdam, VAETrainer
    torch.manual_seed(0)
    model, _ = build_model(TINY, 3, "fp32")
    data = O.synthetic_smiles(64, seed=2)
    kw = dict(n_batch=32, n_epoch=4, kl_start=0, kl_w_start=0.0, kl_w_end=0.1, lr_start=3e-3, lr_end=1e-4, lr_n_period=4,
              lr_n_restarts=1, save_frequency=1, save_train_state=True)
    made = []
    orig = VAETrainer._make_optimizer

    def spy(self):
        made.append(orig(self))
        return made[-1]
    VAETrainer._make_optimizer = spy
    try:
        tr = VAETrainer(model, model_save=str(tmp_path / "a.pt"), log_file=str(tmp_path / "a.csv"), **kw)
        tr.fit(list(data), n_epoch=2)
        st = torch.load(tmp_path / "a_001.state.pt", weights_only=True)
        assert st["epoch"] == 1 and st["optimizer"]["n_steps"] == 4
        assert torch.equal(st["optimizer"]["exp_avg"], made[0].exp_avg.cpu())
        m2 = load_model(VAE, str(tmp_path / "a_001.pt"), "cuda", model_params=dict(q_d_h=32, d_z=16, d_d_h=64, d_n_layers=3))
        tr2 = VAETrainer(m2, model_save=str(tmp_path / "b.pt"), log_file=str(tmp_path / "b.csv"), **kw)
        tr2.fit(list(data), n_epoch=4, resume_from=str(tmp_path / "a_001.state.pt"))
    finally:
        VAETrainer._make_optimizer = orig
    assert made[1].n_steps == 8                      # 4 restored + 2 epochs x 2 batches
    rows = [l.strip().split(",") for l in open(tmp_path / "b.csv") if l.strip().endswith("Train")]
    assert [int(r[0]) for r in rows] == [2, 3]
    # epoch 2 ran at the learning rate the uninterrupted schedule has there
    import math
    lr2 = 1e-4 + (3e-3 - 1e-4) * (1 + math.cos(math.pi * 3 / 4)) / 2   # the scheduler's constructor already stepped once
    assert abs(float(rows[0][2]) - lr2) <= 1e-9 + 1e-6 * lr2
    assert os.path.exists(tmp_path / "b_003.state.pt")


def test_cli_train_then_generate_loop_end_to_end(tmp_path):
    """configs[4] in miniature on the device: `polygon train` through the CLI entry point (run.py:576-620) at the default
    model dims in bf16, then the generate loop (vae_generator.py:82-222) on the trained checkpoint with a stub host
    objective and an identity canonicaliser (RDKit scoring is host-side and out of scope): sample -> dedupe -> score ->
    keep_top -> 75/25 split -> fine-tune, twice."""
    from polygon_amd import run
    from polygon_amd.utils.utils import load_model
    from polygon_amd.vae.vae_generator import SmilesVaeMoleculeGenerator
    from polygon_amd.vae.vae_model import VAE
    data = tmp_path / "train.smiles"
    data.write_text("\n".join(O.synthetic_smiles(384, seed=8)) + "\n")
    ckpt = tmp_path / "cli.pt"
    run.main(["train", "--train_data", str(data), "--model_save", str(ckpt), "--n_epoch", "2", "--n_batch", "128",
              "--device", "cuda:0", "--compute_dtype", "bf16", "--log_file", str(tmp_path / "log.csv"),
              "--save_frequency", "1", "--kl_start", "0", "--kl_w_start", "0.0", "--kl_w_end", "0.05", "--quiet"])
    rows = [l.strip().split(",") for l in open(tmp_path / "log.csv") if l.strip().endswith("Train")]
    assert len(rows) == 2 and float(rows[1][4]) < float(rows[0][4])           # reconstruction loss went down
    assert ckpt.exists() and (tmp_path / "cli_000.pt").exists() and (tmp_path / "cli_001.pt").exists()
    model = load_model(VAE, str(ckpt), "cuda:0", model_params=dict(compute_dtype="bf16"))
    assert len(model.state_dict()) == 74

    class Obj:
        def __init__(self):
            self.calls = 0

        def score_list(self, smiles):
            self.calls += 1
            return [min(len(s), 30) / 30.0 for s in smiles]
    obj = Obj()
    out = tmp_path / "gen"
    out.mkdir()
    gen = SmilesVaeMoleculeGenerator(model, max_len=100, device=torch.device("cuda:0"), out_dir=str(out),
                                     canonicalizer=lambda s, include_stereocenters=True: [x for x in s if x])
    res = gen.optimise(obj, [], keep_top=256, n_epochs=2, mols_to_sample=1024, optimize_n_epochs=2,
                       optimize_batch_size=512, pretrain_n_epochs=1, save_frequency=1, save_payloads=True)
    assert obj.calls == 3 and 0 < len(res) <= 512 and res == sorted(res, reverse=True)
    assert all(0.0 <= r.score <= 1.0 for r in res)
    for f in ("GDM_1.pt", "GDM_2.pt", "GDM_1_payload.txt", "GDM_ongoing_top_scoring_molecules.txt"):
        assert (out / f).exists(), f
    # (no line count: the alphabet holds "\n" and a barely trained model emits it)


def test_adam_leaves_the_gradient_buffer_zeroed_and_the_engine_knows():
    """pg_clip_adam_zero (the loop's zero_grad(), vae_trainer.py:133, folded into the update): after FusedClipAdam.step the
    flat gradient buffer is all zeros, the next fused step skips its own fill and still produces the gradients of a step
    that starts from an explicitly zeroed buffer; a buffer the optimiser has not cleaned is still zeroed by the step."""
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    model, _ = build_model(TINY, 1, "fp32")
    model.train()
    batch = model.get_collate_fn()(list(O.synthetic_smiles(20, seed=4)))
    eps = torch.randn(20, TINY["Dz"], generator=torch.Generator().manual_seed(0)).cuda()
    opt = FusedClipAdam(model, lr=1e-3)
    eng = model._engine
    eng.train_step(batch, 0.3, opt.flat_grad, eps=eps)
    assert float(opt.flat_grad.abs().max()) > 0
    opt.step()
    assert float(opt.flat_grad.abs().max()) == 0.0 and eng.clean_grads is not None
    eng.train_step(batch, 0.3, opt.flat_grad, eps=eps)            # relies on the clean buffer
    assert eng.clean_grads is None
    got = opt.flat_grad.clone()
    other = torch.full_like(opt.flat_grad, 7.0)                   # a dirty buffer of the caller's: zeroed by the step
    eng.train_step(batch, 0.3, other, eps=eps)
    assert rel_err(other, got) <= 1e-6
    # a step on a different stream than the one the optimiser ran on does not trust the marker
    opt.flat_grad.copy_(other)
    opt.step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        opt.flat_grad.fill_(3.0)                                   # (somebody dirtied it meanwhile, on that stream)
        eng.train_step(batch, 0.3, opt.flat_grad, eps=eps)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    ref = got.clone()                                             # parameters moved by two updates of lr 1e-3: close, not equal
    assert rel_err(opt.flat_grad, ref) <= 0.2, rel_err(opt.flat_grad, ref)     # (a buffer left at 3.0 + gradient would be off by orders of magnitude)


def test_side_stream_prologue_and_token_sort_follow_changing_batches():
    """The fused step under deferral packs the decoder's batch, clears its counters and sorts the rows by token on the
    library's side stream, beside the encoder forward (pg_vae.hip: note_tokens / side_pro).  Batches that change size and
    content from step to step (new workspace, new tokens behind the same entry points) must give the gradients of the
    same step run entirely in order (deferral off: everything on the caller's stream, the rows sorted in the backward
    pass).  Differences are the split-K atomics' summation order only."""
    import ctypes
    from polygon_amd import _lib
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    L = _lib.lib()
    model, _ = build_model(REAL, 11, "bf16", d_dropout=0.2)
    model.train()
    eng = model._engine
    opt = FusedClipAdam(model, lr=1e-3)
    ref_grad = torch.zeros_like(opt.flat_grad)
    for step, (n, seed) in enumerate([(300, 1), (300, 2), (64, 3), (700, 4), (300, 5)]):
        batch = model.get_collate_fn()(list(O.synthetic_smiles(n, seed=seed)))
        eps = torch.from_numpy(np.random.default_rng(seed).standard_normal((n, REAL["Dz"])).astype(np.float32)).cuda()
        # the shipped path first: the weights changed in the last opt.step(), so this call rebuilds the shadows on the side
        # stream and the prologue + sorts follow them there
        kl1, rec1 = eng.train_step(batch, 0.3, opt.flat_grad, eps=eps, p_drop=0.2, seed=77 + step)
        torch.cuda.synchronize()
        # in order, on the same weights (the shadows are current: nothing is pending on the side stream)
        ref_grad.zero_()
        old = L.pg_defer_weight_grads(0)
        try:
            kl0, rec0 = eng._train_step(batch, 0.3, ref_grad, eps, None, 0.2, 77 + step, None, 0.0, 0)
        finally:
            L.pg_defer_weight_grads(old)
        torch.cuda.synchronize()
        assert abs(kl0.item() - kl1.item()) <= 1e-5 * abs(kl0.item()) and abs(rec0.item() - rec1.item()) <= 1e-5 * abs(rec0.item())
        err = ((opt.flat_grad - ref_grad).abs().max() / ref_grad.abs().max()).item()
        assert err <= 2e-3, (step, err)
        opt.step()                                   # new weights: the next step rebuilds the shadows on the side stream
    _lib.check(L.pg_check_failure(_lib.stream_ptr()), "steps")
