"""Development aid: wall-clock stamps (100 MHz) of one workgroup of the group-persistent decoder kernel, steps 8..15."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polygon_amd import _lib
from polygon_amd.utils.synthetic import synthetic_smiles
from polygon_amd.vae.vae_model import VAE
from polygon_amd.vae.vae_trainer import FusedClipAdam

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 1
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # 0 forward kernel, 1 BPTT kernel
first = int(sys.argv[3]) if len(sys.argv) > 3 else 8    # first of the 8 traced steps (BPTT counts steps from the end)
group = int(sys.argv[4]) if len(sys.argv) > 4 else 0    # traced group (8, 9: the groups whose layers sit on different XCDs)
torch.manual_seed(0)
model = VAE(d_dropout=0.2, compute_dtype="bf16").cuda().train()
opt = FusedClipAdam(model, lr=3e-4, clip_grad=50)
eng = model._engine
eng.ensure_shadow()
gb = model.get_collate_fn()(synthetic_smiles(1024, seed=1))
eps = torch.randn(1024, model.d_z, device="cuda")
stamps = torch.zeros(8 + 64, dtype=torch.int64, device="cuda")
stamps[0] = layer
stamps[1] = first
stamps[2] = group
L = _lib.lib()
for i in range(3):
    eng.train_step(gb, 0.5, opt.flat_grad, eps=eps, p_drop=0.2, seed=i)
torch.cuda.synchronize()
_lib.check(L.pg_persist_stamps(_lib.ptr(stamps), which))
eng.train_step(gb, 0.5, opt.flat_grad, eps=eps, p_drop=0.2, seed=7)
torch.cuda.synchronize()
_lib.check(L.pg_persist_stamps(None, 0))
s = stamps.cpu().numpy()[8:].reshape(8, 8)
names = ["step start", "polls done", "fence+barrier", "tile0 staged", "tile0 done", "all tiles", "drained", "arrived"]
if which == 1:
    names[3] = "ring primed"
print(("BPTT " if which else "forward ") + "layer %d group %d, us since previous stamp (10 ns ticks):" % (layer, group))
for k in range(8):
    row = s[k]
    print("s=%2d " % (first + k) + "  ".join("%s %.2f" % (names[i], (row[i] - (row[i - 1] if i else s[k - 1][7] if k else row[0])) / 100.0) for i in range(8)))
