"""Development aid: per-tensor gradient deviation of the persistent kernels from the per-step kernels (bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import vae_oracle as O
from polygon_amd import _lib
from tests.helpers import REAL, build_model, rel_err
from polygon_amd.vae.vae_trainer import FusedClipAdam
n, p_drop = int(sys.argv[1]) if len(sys.argv) > 1 else 300, 0.2
model, _ = build_model(REAL, 21, "bf16", d_dropout=p_drop)
model.train()
strings = list(O.synthetic_smiles(n, seed=n))
batch = model.get_collate_fn()(strings)
eps = torch.from_numpy(np.random.default_rng(n).standard_normal((n, REAL["Dz"])).astype(np.float32)).cuda()
opt = FusedClipAdam(model, lr=0.0)
L = _lib.lib()
def run(mask):
    old = L.pg_persist_disable(mask)
    kl, rec = model._engine.train_step(batch, 0.5, opt.flat_grad, eps=eps, p_drop=p_drop, seed=4242)
    torch.cuda.synchronize(); L.pg_persist_disable(old)
    return kl.item(), rec.item(), opt.flat_grad.clone()
ref = run(7)
for mask in (7, 5, 5, 3, 0, 0):
    got = run(mask)
    off, errs = 0, []
    for p in model._engine.params():
        k = p.numel(); errs.append(rel_err(got[2][off:off + k], ref[2][off:off + k])); off += k
    print("mask %d  d_kl %.2e  d_rec %.2e  max grad rel err %.4f (tensor %d)  all: %s" % (
        mask, abs(got[0] - ref[0]) / abs(ref[0]), abs(got[1] - ref[1]) / abs(ref[1]), max(errs), int(np.argmax(errs)),
        " ".join("%.3f" % e for e in errs)))
