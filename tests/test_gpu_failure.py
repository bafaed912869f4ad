"""A persistent recurrence kernel that gives up (its peer workgroups never arrive) must not corrupt anything: the Adam
kernel skips the update on the device, the host reports PG_ERR_PERSIST at its next check - before anything is saved -
and later calls run on the per-step kernels.  The give-up is a REAL one: pg_persist_spin_limit(0) makes every unsatisfied
wait of the kernels' bounded polling loops expire at once."""
import numpy as np
import pytest
import torch

from oracle import vae_oracle as O
from polygon_amd import _lib
from tests.helpers import REAL, build_model

pytestmark = pytest.mark.gpu


def test_persistent_give_up_skips_update_raises_and_falls_back():
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    L = _lib.lib()
    model, _ = build_model(REAL, 3, "bf16", d_dropout=0.0)
    model.train()
    eng = model._engine
    opt = FusedClipAdam(model, lr=1e-2)
    n = 200
    batch = model.get_collate_fn()(list(O.synthetic_smiles(n, seed=1)))
    eps = torch.from_numpy(np.random.default_rng(0).standard_normal((n, REAL["Dz"])).astype(np.float32)).cuda()
    assert L.pg_persist_disable(-1) == 0
    # a healthy step first (also allocates the give-up words)
    eng.train_step(batch, 0.5, opt.flat_grad, eps=eps)
    opt.step()
    _lib.check(L.pg_check_failure(_lib.stream_ptr()), "healthy step")
    before = eng.flat.clone()
    m_before = opt.exp_avg.clone()
    import ctypes
    L.pg_persist_spin_limit.restype = ctypes.c_uint
    old = L.pg_persist_spin_limit(ctypes.c_long(0))
    try:
        eng.train_step(batch, 0.5, opt.flat_grad, eps=eps)      # decoder layers > 0 give up at their first wait
        opt.step()
        torch.cuda.synchronize()
    finally:
        L.pg_persist_spin_limit(ctypes.c_long(old))
    assert torch.equal(eng.flat, before), "the optimiser applied gradients of a failed recurrence"
    assert torch.equal(opt.exp_avg, m_before)
    # the next check reports it, once
    st = L.pg_check_failure(_lib.stream_ptr())
    assert st == -5 and b"persistent recurrence kernel timed out" in L.pg_last_error()
    assert L.pg_check_failure(_lib.stream_ptr()) == 0
    assert L.pg_persist_disable(-1) == 15                       # every recurrence now on the per-step kernels
    try:
        kl, rec = eng.train_step(batch, 0.5, opt.flat_grad, eps=eps)
        g_step = opt.flat_grad.clone()
        opt.step()
        torch.cuda.synchronize()
        assert np.isfinite(kl.item()) and np.isfinite(rec.item())
        assert not torch.equal(eng.flat, before)                # updates are applied again
    finally:
        L.pg_persist_disable(0)
    # and the persistent kernels still work when re-enabled: same gradients as the per-step ones on the same weights
    opt2 = FusedClipAdam(model, lr=0.0)
    L.pg_persist_disable(15)
    try:
        eng.train_step(batch, 0.5, opt2.flat_grad, eps=eps)
        ref = opt2.flat_grad.clone()
    finally:
        L.pg_persist_disable(0)
    eng.train_step(batch, 0.5, opt2.flat_grad, eps=eps)
    _lib.check(L.pg_check_failure(_lib.stream_ptr()), "re-enabled")
    err = ((opt2.flat_grad - ref).abs().max() / ref.abs().max()).item()
    assert err <= 5e-3, err


def test_trainer_raises_before_saving(tmp_path):
    """VAETrainer.fit: a give-up during the epoch surfaces as PolygonHipError after the epoch's sync and BEFORE the
    checkpoint of that epoch is written."""
    import ctypes
    from polygon_amd.vae.vae_trainer import VAETrainer
    L = _lib.lib()
    L.pg_persist_spin_limit.restype = ctypes.c_uint
    model, _ = build_model(REAL, 3, "bf16", d_dropout=0.0)
    save = tmp_path / "m.pt"
    tr = VAETrainer(model, n_batch=64, n_epoch=1, kl_start=0, model_save=str(save), prefetch=0)
    old = L.pg_persist_spin_limit(ctypes.c_long(0))
    try:
        with pytest.raises(_lib.PolygonHipError):
            tr.fit(list(O.synthetic_smiles(64, seed=2)))
    finally:
        L.pg_persist_spin_limit(ctypes.c_long(old))
        L.pg_persist_disable(0)
    assert not list(tmp_path.glob("*.pt")), "a checkpoint was written after a failed recurrence"


def test_layout_of_saved_activations_is_recorded_not_rederived():
    """The forward pass records, per workspace, whether it saved r/z/n/hn in lane order (for the persistent BPTT) or as
    packed rows; pg_decoder_bwd reads that record.  Switching the persistent BPTT off BETWEEN the two calls (what a
    detected give-up does) must fail loudly - the per-step BPTT cannot read lane order - and a step restarted from the
    forward pass under the new mask must give the gradients of an undisturbed step."""
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    L = _lib.lib()
    model, _ = build_model(REAL, 5, "bf16", d_dropout=0.0)
    model.train()
    eng = model._engine
    opt = FusedClipAdam(model, lr=0.0)
    n = 96
    batch = model.get_collate_fn()(list(O.synthetic_smiles(n, seed=4)))
    eps = torch.from_numpy(np.random.default_rng(1).standard_normal((n, REAL["Dz"])).astype(np.float32)).cuda()
    assert L.pg_persist_disable(-1) == 0
    eng.train_step(batch, 0.5, opt.flat_grad, eps=eps)
    ref = opt.flat_grad.clone()
    ws = eng._ws
    z, dz, scal = eng._small
    # forward with the persistent kernels (lane-order saves), then the BPTT kernel is switched off
    eng.encoder_fwd(ws, batch, eps, None, 1, z, None, None, scal[0:1])
    eng.decoder_fwd(ws, batch, z, None, 0.0, 0, 0, 1, scal[1:2])
    L.pg_persist_disable(4)
    try:
        with pytest.raises(_lib.PolygonHipError, match="restart the step from pg_decoder_fwd"):
            eng.decoder_bwd(ws, batch, None, 0.0, 0, 0, 1.0, None, opt.flat_grad, dz)
        # restarted from the forward pass under the new mask: packed rows + per-step BPTT
        eng.train_step(batch, 0.5, opt.flat_grad, eps=eps)
        err = ((opt.flat_grad - ref).abs().max() / ref.abs().max()).item()
        assert err <= 5e-3, err
        # and the other direction: packed rows saved under mask 4 are read as packed rows by the persistent BPTT
        eng.encoder_fwd(ws, batch, eps, None, 1, z, None, None, scal[0:1])
        eng.decoder_fwd(ws, batch, z, None, 0.0, 0, 0, 1, scal[1:2])
    finally:
        L.pg_persist_disable(0)
    opt.flat_grad.zero_()
    eng.decoder_bwd(ws, batch, None, 0.0, 0, 0, 1.0, None, opt.flat_grad, dz)
    eng.encoder_bwd(ws, batch, None, dz, 0.5, None, opt.flat_grad)
    torch.cuda.synchronize()
    err = ((opt.flat_grad - ref).abs().max() / ref.abs().max()).item()
    assert err <= 5e-3, err
