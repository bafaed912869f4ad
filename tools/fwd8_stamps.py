"""Development aid: cycle stamps of one wave pair of the two-waves-per-SIMD forward kernel (library built with
-DPG_FWD8_KNOCKOUTS, run with PG_FWD8_KO=32): per tile iteration of step 20, cycles since the iteration's start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polygon_amd import _lib
from polygon_amd.utils.synthetic import synthetic_smiles
from polygon_amd.vae.vae_model import VAE
from polygon_amd.vae.vae_trainer import FusedClipAdam
torch.manual_seed(0)
model = VAE(d_dropout=0.2, compute_dtype="bf16").cuda().train()
opt = FusedClipAdam(model, lr=3e-4, clip_grad=50)
eng = model._engine
eng.ensure_shadow()
gb = model.get_collate_fn()(synthetic_smiles(1024, seed=1))
eps = torch.randn(1024, model.d_z, device="cuda")
stamps = torch.zeros(8 + 128 + 64, dtype=torch.int64, device="cuda")
L = _lib.lib()
for i in range(3):
    eng.train_step(gb, 0.5, opt.flat_grad, eps=eps, p_drop=0.2, seed=i)
torch.cuda.synchronize()
_lib.check(L.pg_persist_stamps(_lib.ptr(stamps), 0))
eng.train_step(gb, 0.5, opt.flat_grad, eps=eps, p_drop=0.2, seed=7)
torch.cuda.synchronize()
_lib.check(L.pg_persist_stamps(None, 0))
raw = stamps.cpu().numpy()
s = raw[8:8 + 128].reshape(2, 8, 8)
st = raw[8 + 128:].reshape(2, 4, 8)
t0 = s[0, 0, 0]
names = [["start", "dma issued", "epilogue", "mfma", "waited"], ["start", "stores", "mfma", "epilogue", "ready"]]
for role in range(2):
    print("role %d (%s side)" % (role, "hx"[role]))
    for i in range(8):
        if s[role, i, 0] == 0:
            continue
        print("  tile %d: " % i + "  ".join("%s %d" % (names[role][k], s[role, i, k] - t0) for k in range(5)))
snames = ["step begin", "polled+BarS", "tile0 landed", "loop done", "E1", "stores drained", "E2"]
for role in range(2):
    print("role %d step-level (cycles since step 20 began on the h side):" % role)
    for k in range(4):
        print("  step %d: " % (20 + k) + "  ".join("%s %d" % (snames[j], st[role, k, j] - st[0, 0, 0]) for j in range(7)))
