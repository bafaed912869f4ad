"""BASELINE.json's full sizes.  DIRECT oracle parity of one training step at configs[0] (n_batch 512, fp32) and configs[1]
(n_batch 1024, bf16, d_dropout 0.2, the persistent kernels in their 10-group configuration; the numpy oracle takes a few
seconds per step at these sizes), the sampler's > 65 536-row chunk loop against separate calls, and size-independent
properties: the sampler at 8192 rows (configs[3]/[4]) and the fused training step at both sizes."""
import ctypes
import numpy as np
import pytest
import torch

from oracle import vae_oracle as O
from tests.helpers import REAL, build_model, dropout_masks, rel_err

pytestmark = pytest.mark.gpu


def test_sampler_8192_rows_properties():
    """Token matrix well-formed; the stream of a row depends on (seed, GLOBAL row) only, so sharding the rows over calls
    (= over GPUs) does not change a single token; the same seed reproduces; equal z rows in argmax mode give
    equal strings."""
    model, _ = build_model(REAL, 31, "bf16", eos_bias=3.0)
    model.eval()
    n, V = 8192, len(model.vocabulary)
    z = torch.randn(n, REAL["Dz"], generator=torch.Generator().manual_seed(5)).cuda()
    tok, ep = model.sample_tokens(n, max_len=100, z=z, seed=77)
    t, e = tok.cpu().numpy().astype(np.int64), ep.cpu().numpy().astype(np.int64)
    assert t.shape == (n, 100) and (t < V).all() and (t[:, 0] == model.bos).all()
    assert ((e >= 2) & (e <= 100)).all()
    cols = np.arange(100)[None, :]
    assert (t[cols >= e[:, None]] == model.pad).all()               # nothing written past a row's end
    is_eos = (t == model.eos) & (cols < e[:, None])
    first = np.where(is_eos.any(1), is_eos.argmax(1), 99)
    assert (e == np.minimum(first + 1, 100)).all()                   # end = one past the first eos (or max_len)
    assert is_eos.any(1).mean() > 0.5                                # the eos bias makes most rows finish: the bookkeeping is exercised
    # sharded = unsharded, bit for bit (shards above 2048 rows: below that the cell runs with split-K groups, another
    # summation order, and a bf16 stream may then differ in rare tokens)
    parts = [model.sample_tokens(4096, max_len=100, z=z[o:o + 4096], seed=77, row_offset=o) for o in range(0, n, 4096)]
    assert torch.equal(torch.cat([p[0] for p in parts]), tok) and torch.equal(torch.cat([p[1] for p in parts]), ep)
    tok2, ep2 = model.sample_tokens(n, max_len=100, z=z, seed=77)
    assert torch.equal(tok2, tok) and torch.equal(ep2, ep)
    tok3, _ = model.sample_tokens(n, max_len=100, z=z, seed=78)
    assert not torch.equal(tok3, tok)
    zz = z[:4].repeat(2048, 1)
    s = model.sample(n, max_len=100, z=zz, multinomial=False)
    assert s[:4] * 2047 == s[4:]


@pytest.mark.parametrize("dt,n,p_drop", [("bf16", 1024, 0.2), ("fp32", 512, 0.0)])
def test_train_step_full_batch_properties(dt, n, p_drop):
    """Gradient is affine in the KL weight (same masks, same eps): g(0.5) = (g(0) + g(1)) / 2; losses do not depend on
    it; a few Adam steps on one batch lower the loss."""
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    model, _ = build_model(REAL, 41, dt, d_dropout=p_drop)
    model.train()
    batch = model.get_collate_fn()(list(O.synthetic_smiles(n, seed=n)))
    eps = torch.randn(n, REAL["Dz"], generator=torch.Generator().manual_seed(3)).cuda()
    opt = FusedClipAdam(model, lr=1e-3)
    eng = model._engine
    out = {}
    for w in (0.0, 1.0, 0.5):
        kl, rec = eng.train_step(batch, w, opt.flat_grad, eps=eps, p_drop=p_drop, seed=9)
        out[w] = (kl.item(), rec.item(), opt.flat_grad.clone())
    assert torch.isfinite(out[0.5][2]).all()
    assert abs(out[0.0][0] - out[1.0][0]) <= 1e-5 * abs(out[1.0][0]) and abs(out[0.0][1] - out[1.0][1]) <= 1e-5 * abs(out[1.0][1])
    mid = 0.5 * (out[0.0][2] + out[1.0][2])
    assert rel_err(out[0.5][2], mid) <= (2e-3 if dt == "bf16" else 1e-4)
    first = out[0.5][1]
    for _ in range(8):
        kl, rec = eng.train_step(batch, 0.5, opt.flat_grad, eps=eps, p_drop=p_drop, seed=9)
        opt.step()
    assert rec.item() < first - 0.05, (first, rec.item())


def _profile_read(which):
    from polygon_amd import _lib
    ms, n, fl, seg = ctypes.c_double(), ctypes.c_long(), ctypes.c_double(), ctypes.c_int()
    _lib.check(_lib.lib().pg_profile_read(which, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(seg)))
    return n.value, seg.value


@pytest.mark.parametrize("dt,n,p_drop", [("fp32", 512, 0.0), ("fp32", 512, 0.2), ("bf16", 1024, 0.2)],
                         ids=["configs0_fp32_512", "configs0_fp32_512_dropout", "configs1_bf16_1024_dropout"])
def test_train_step_full_batch_matches_oracle(dt, n, p_drop):
    """One training step at BASELINE's batch sizes against the numpy oracle directly (vae_model.py:151-166, 216-275 and
    their autograd): kl / recon within 1e-3 (fp32) or 2e-2 (bf16), all 25 gradients within 1e-2.  With dropout the oracle
    takes the kernels' own masks (pg_dropout_masks_host, float32 to bound memory).  The bf16 case must have run the
    group-persistent recurrences (one launch each), which at 1024 rows means their 10-group / 240-workgroup layout."""
    from polygon_amd import _lib
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    seed = 20250310
    model, P = build_model(REAL, 23, dt, d_dropout=p_drop)
    model.train()
    strings = list(O.synthetic_smiles(n, seed=n + 1))
    batch = model.get_collate_fn()(list(strings))
    eps = np.random.default_rng(11).standard_normal((n, REAL["Dz"])).astype(np.float32)
    opt = FusedClipAdam(model, lr=0.0)
    L = _lib.lib()
    _lib.check(L.pg_profile_enable(1))
    try:
        _profile_read(0), _profile_read(1)
        kl, rec = model._engine.train_step(batch, 0.5, opt.flat_grad, eps=torch.from_numpy(eps).cuda(), p_drop=p_drop, seed=seed)
        torch.cuda.synchronize()
        (nf, sf), (nb, sb) = _profile_read(0), _profile_read(1)
    finally:
        _lib.check(L.pg_profile_enable(0))
    if dt == "bf16":
        assert nf == sf == 1 and nb == sb == 1, ("persistent recurrences did not run", nf, sf, nb, sb)
    else:
        assert nf > 1 and nb > 1                                   # fp32: one launch per anti-diagonal
    seqs = O.collate(strings)
    T = max(len(s) for s in seqs)
    dm = dropout_masks(seed, p_drop, REAL["L"] - 1, n, T, REAL["Hd"], dtype=np.float32) if p_drop > 0 else None
    enc, dec = O.vae_forward(P, seqs, eps.astype(np.float64), dm)
    G, _ = O.vae_backward(P, enc, dec, g_kl=0.5, g_recon=1.0)
    tol = 1e-3 if dt == "fp32" else 2e-2
    assert abs(kl.item() - enc["kl"]) <= tol * abs(enc["kl"]), (kl.item(), enc["kl"])
    assert abs(rec.item() - dec["recon"]) <= tol * abs(dec["recon"]), (rec.item(), dec["recon"])
    worst = 0.0
    for name, g in zip(O.param_order(P), model._engine.grad_views(opt.flat_grad)):
        e = rel_err(g, G[name])
        worst = max(worst, e)
        assert e <= 1e-2, (name, e)
    print("n_batch %d %s p_drop %.1f: kl %.3e recon %.3e worst gradient %.2e" % (
        n, dt, p_drop, abs(kl.item() - enc["kl"]) / abs(enc["kl"]), abs(rec.item() - dec["recon"]) / abs(dec["recon"]), worst))


def test_sampler_chunk_loop_equals_separate_calls():
    """n_batch above the 65 536 rows one library call takes: VAE.sample_tokens walks chunks (uniforms[:, r0:r0+m],
    row_offset + r0; vae_model.py sample_tokens).  70 000 rows with pre-drawn uniforms must equal two separate calls on
    the two row ranges, and - Philox mode - two calls that pass the chunks' row offsets."""
    model, _ = build_model(REAL, 31, "bf16", eos_bias=1.0)
    model.eval()
    n, CH, max_len = 70000, 65536, 8
    g = torch.Generator().manual_seed(9)
    z = torch.randn(n, REAL["Dz"], generator=g).cuda()
    u = torch.rand(max_len - 1, n, generator=g).cuda()
    tok, ep = model.sample_tokens(n, max_len=max_len, z=z, uniforms=u)
    a = model.sample_tokens(CH, max_len=max_len, z=z[:CH], uniforms=u[:, :CH].contiguous())
    b = model.sample_tokens(n - CH, max_len=max_len, z=z[CH:], uniforms=u[:, CH:].contiguous(), row_offset=CH)
    assert torch.equal(tok, torch.cat([a[0], b[0]])) and torch.equal(ep, torch.cat([a[1], b[1]]))
    t = tok.cpu().numpy()
    assert (t[:, 0] == model.bos).all() and len(np.unique(t[:, 1])) > 10          # rows really differ
    tok2, ep2 = model.sample_tokens(n, max_len=max_len, z=z, seed=5, row_offset=100)
    a = model.sample_tokens(CH, max_len=max_len, z=z[:CH], seed=5, row_offset=100)
    b = model.sample_tokens(n - CH, max_len=max_len, z=z[CH:], seed=5, row_offset=100 + CH)
    assert torch.equal(tok2, torch.cat([a[0], b[0]])) and torch.equal(ep2, torch.cat([a[1], b[1]]))
    assert not torch.equal(tok2, tok)


@pytest.mark.parametrize("generic", [False, True], ids=["default_encoder", "bidir_encoder"])
def test_batches_above_the_persistent_row_limit_run_as_micro_batches(generic):
    """n_batch 2500 > 1024 rows (dec_persist_supported): the fused bf16 step splits the batch into 3 interleaved
    micro-batches that all run the one-launch recurrences and accumulate into one gradient buffer.  Checked against the
    per-step kernels on the whole batch (pg_persist_disable(7), which take any B): same losses and gradients, dropout on
    (the masks are keyed by the row's index in the whole batch, so the split does not move them)."""
    import ctypes
    from polygon_amd import _lib
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    dims = dict(REAL, Lq=2, bidir=True) if generic else REAL
    model, _ = build_model(dims, 31, "bf16", d_dropout=0.2, q_dropout=0.1)
    model.train()
    n = 2500
    batch = model.get_collate_fn()(list(O.synthetic_smiles(n, seed=77)))
    eps = torch.randn(n, REAL["Dz"], generator=torch.Generator().manual_seed(5)).cuda()
    opt = FusedClipAdam(model, lr=0.0)
    eng = model._engine
    assert eng.micro_batches(n) == 3 and eng.micro_batches(1024) == 1
    kw = dict(p_drop=0.2, seed=4242, q_drop=0.1 if generic else 0.0, q_seed=99)
    L = _lib.lib()
    L.pg_profile_enable(1)
    try:
        kl, rec = [x.item() for x in eng.train_step(batch, 0.5, opt.flat_grad, eps=eps, **kw)]
        torch.cuda.synchronize()
        ms, launches, flop, seg = ctypes.c_double(), ctypes.c_long(), ctypes.c_double(), ctypes.c_int()
        _lib.check(L.pg_profile_read(0, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(flop), ctypes.byref(seg)), "profile")
        assert (seg.value, launches.value) == (3, 3), (seg.value, launches.value)      # three one-launch forward recurrences
        _lib.check(L.pg_profile_read(1, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(flop), ctypes.byref(seg)), "profile")
        assert (seg.value, launches.value) == (3, 3), (seg.value, launches.value)      # ... and three one-launch BPTTs
    finally:
        L.pg_profile_enable(0)
    got = opt.flat_grad.clone()
    old = L.pg_persist_disable(7)
    try:
        eng.micro_batches = lambda B: 1
        kl0, rec0 = [x.item() for x in eng.train_step(batch, 0.5, opt.flat_grad, eps=eps, **kw)]
        torch.cuda.synchronize()
    finally:
        L.pg_persist_disable(old)
        del eng.micro_batches
    assert abs(kl - kl0) <= 2e-3 * abs(kl0) and abs(rec - rec0) <= 2e-3 * abs(rec0), (kl, kl0, rec, rec0)
    worst = max((rel_err(a, b), name) for a, b, (name, _) in
                zip(eng.grad_views(got), eng.grad_views(opt.flat_grad), model.vae.named_parameters()))
    assert worst[0] <= 1e-2, worst
    # forward-only (validation) path: same split, same sums
    model.eval()
    k1, r1 = [x.item() for x in eng.eval_losses(batch, eps=eps)]
    assert abs(k1 - kl0) <= 2e-2 * abs(kl0)        # (train mode above had q_dropout on for the bidirectional encoder)
