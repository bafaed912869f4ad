"""Development aid: two gloo ranks on one GPU; rank 1's persistent kernels give up; prints what each rank sees."""
import os, subprocess, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "RANK" not in os.environ:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0",
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for r in range(2)]
    sys.exit(max(p.wait() for p in ps))
sys.path.insert(0, ROOT)
import ctypes
import numpy as np, torch, torch.distributed as dist
from polygon_amd.parallel import init_distributed, GradComm
from polygon_amd.vae.vae_trainer import FusedClipAdam
from polygon_amd import _lib
from oracle import vae_oracle as O
from tests.helpers import REAL, build_model
rank, world, local = init_distributed(backend="gloo")
torch.cuda.set_device(0)
L = _lib.lib()
L.pg_persist_spin_limit.restype = ctypes.c_uint
model, _ = build_model(REAL, 21, "bf16", device="cuda:0", d_dropout=0.0)
model.train()
eng = model._engine
opt = FusedClipAdam(model, lr=1e-2)
strings = O.synthetic_smiles(64, seed=6)
sh = model.get_collate_fn(shard=(rank, world))(list(strings))
eps = torch.from_numpy(np.random.default_rng(2).standard_normal((64, REAL["Dz"])).astype(np.float32)).cuda()[rank::world].contiguous()
comm = GradComm()
eng.train_step(sh, 0.4, opt.flat_grad, eps=eps, comm=comm); opt.step()
print(rank, "healthy check", L.pg_check_failure(_lib.stream_ptr()), flush=True)
before = eng.flat.clone()
old = L.pg_persist_spin_limit(ctypes.c_long(0)) if rank == 1 else None
checks = L.pg_entry_checks(0)
eng.train_step(sh, 0.4, opt.flat_grad, eps=eps, comm=comm)
print(rank, "flag after step", comm._flag.item(), "persist mask", L.pg_persist_disable(-1), flush=True)
f = torch.zeros(1, dtype=torch.int32, device="cuda")
_lib.check(L.pg_fail_flag_export(_lib.ptr(f), _lib.stream_ptr()))
print(rank, "local word now", f.item(), flush=True)
opt.step()
torch.cuda.synchronize()
L.pg_entry_checks(checks)
print(rank, "equal", torch.equal(eng.flat, before), "check", L.pg_check_failure(_lib.stream_ptr()), flush=True)
dist.barrier(); dist.destroy_process_group()
