"""Measures (prints) the errors the bf16 / dropout parity tests pin: run on the GPU box, numbers go into the tests."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vae_oracle as O  # noqa: E402
from polygon_amd import _lib  # noqa: E402
from tests.helpers import REAL, build_model, canon_named, rel_err, dropout_masks  # noqa: E402
from polygon_amd.vae.vae_trainer import FusedClipAdam  # noqa: E402


def grads_vs_oracle(dt, p_drop, mask, n=48, seed=4242):
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
    dm = dropout_masks(seed, p_drop, REAL["L"] - 1, n, T, REAL["Hd"]) if p_drop > 0 else None
    enc, dec = O.vae_forward(P, seqs, eps.astype(np.float64), dm)
    G, _ = O.vae_backward(P, enc, dec, g_kl=0.5, g_recon=1.0)
    gv = model._engine.grad_views(opt.flat_grad)
    errs = {}
    for name, g in zip(O.param_order(P), gv):
        errs[name] = rel_err(g, G[name])
    print("dt=%s p=%.1f mask=%d  kl %.2e recon %.2e  grad max %.2e (%s)" % (
        dt, p_drop, mask, abs(kl.item() - enc["kl"]) / abs(enc["kl"]), abs(rec.item() - dec["recon"]) / abs(dec["recon"]),
        max(errs.values()), max(errs, key=errs.get)))
    return errs


if __name__ == "__main__":
    for dt in ("fp32", "bf16"):
        for p in (0.0, 0.2):
            for mask in ((0, 7) if dt == "bf16" else (0,)):
                e = grads_vs_oracle(dt, p, mask)
                if dt == "bf16":
                    print("   ", {k: "%.1e" % v for k, v in e.items()})
