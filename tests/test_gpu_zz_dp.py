"""Data-parallel equivalence on ONE GPU.

* two ranks (gloo, both on cuda:0) each tokenise and run the fused step on THEIR shard of a global batch (sharding
  collate, global normalisers, dropout keyed by global row, deferred weight gradients joined before the reduction); the
  all-reduced flat gradient and the summed losses must equal the single-process step on the whole batch; the trainer's
  logged losses must be the global-batch values, not a rank's share;
* the two shards run in ONE process through the persistent kernels and add up to the whole batch;
* world_size 1 on backend nccl: the real GradComm (RCCL all-reduce on its own stream, overlapped with the encoder BPTT,
  deferred weight gradients joined on the communication stream) executes on hardware beside the persistent kernels."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, faulthandler
faulthandler.dump_traceback_later(110, exit=False)   # if a rank stalls, its stack goes to stderr before the parent gives up
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from polygon_amd.parallel import init_distributed, GradComm
from polygon_amd.vae.vae_trainer import FusedClipAdam, VAETrainer
from oracle import vae_oracle as O
from tests.helpers import REAL, build_model
rank, world, local = init_distributed(backend="gloo")
torch.cuda.set_device(0)
print("rank", rank, "init ok", flush=True)

for dt in ("fp32", "bf16"):
    model, _ = build_model(REAL, 21, dt, device="cuda:0", d_dropout=0.2)
    model.train()
    eng = model._engine
    opt = FusedClipAdam(model, lr=3e-4)
    strings = O.synthetic_smiles(70, seed=4) + ["C", ""]
    gb = model.get_collate_fn()(list(strings))
    B = len(gb)
    eps = torch.from_numpy(np.random.default_rng(1).standard_normal((B, REAL["Dz"])).astype(np.float32)).cuda()
    # single-process reference on the whole batch
    kl0, rec0 = eng.train_step(gb, 0.4, opt.flat_grad, eps=eps, p_drop=0.2, seed=99)
    g0 = opt.flat_grad.clone(); kl0 = kl0.item(); rec0 = rec0.item()
    # sharded step: this rank tokenises only its rows (the collate sorts the global batch on lengths first)
    sh = model.get_collate_fn(shard=(rank, world))(list(strings))
    assert len(sh) == len(range(rank, B, world)) and sh.dp is not None and sh.global_rows == B
    comm = GradComm()                            # gloo: reduced through host copies, after joining the deferred GEMMs
    kl, rec = eng.train_step(sh, 0.4, opt.flat_grad, eps=eps[rank::world].contiguous(), p_drop=0.2, seed=99, comm=comm)
    losses = comm.sum_scalars(torch.stack([kl, rec]))   # per-rank shares (globally normalised) sum to the global loss
    g = opt.flat_grad
    err = ((g - g0).abs().max() / g0.abs().max()).item()
    tol = 1e-4 if dt == "fp32" else 2e-2
    assert err <= tol, (dt, "grad", err)
    assert abs(losses[0].item() - kl0) <= 1e-4 * abs(kl0) + 1e-7, (dt, losses[0].item(), kl0)
    assert abs(losses[1].item() - rec0) <= (1e-4 if dt == "fp32" else 5e-3) * abs(rec0), (dt, losses[1].item(), rec0)
    print("rank", rank, dt, "ok grad_rel_err=%%.2e" %% err, flush=True)

# trainer level: different initial weights per rank are overwritten by rank 0's (broadcast in fit); the logged losses are
# the GLOBAL-batch values (a rank's share of the recon loss would be ~ log(51) / 2)
torch.manual_seed(100 + rank)
model, _ = build_model(REAL, 30 + rank, "bf16", device="cuda:0", d_dropout=0.2)
log = os.path.join(%r, "dp_log_%%d.csv" %% rank)
tr = VAETrainer(model, n_batch=24, n_epoch=1, kl_start=0, kl_w_start=0.1, model_save=None, log_file=log, lr_start=1e-5, lr_end=1e-5)
data = O.synthetic_smiles(97, seed=8)            # 2 global batches of 48 + a tail of ONE string: skipped on both ranks
tr.fit(list(data), val_data=list(O.synthetic_smiles(50, seed=9)))
flat = model._engine.flat.clone()
other = flat.cpu(); dist.broadcast(other, 0)
assert torch.equal(other, flat.cpu()), "ranks ended with different weights"
if rank == 0:
    rows = [l.strip().split(",") for l in open(log) if l.strip().endswith(("Train", "Eval"))]
    assert [r[-1] for r in rows] == ["Train", "Eval"], rows
    for r in rows:
        recon = float(r[4])
        n_b = 3 if r[-1] == "Train" else 2       # CircularBuffer.mean() divides by the number of loader batches
        got = recon * n_b / 2.0                   # two batches carried a loss in either pass
        assert 3.6 < got < 4.3, (r, got)          # ~ log(51) = 3.93 for a near-untrained model; a rank's share would be ~1.97
print("rank", rank, "trainer ok", flush=True)

# sharded sampling: rows split over the ranks, one final gather; same strings as ONE process decoding every row
model.eval()
n, ml, seed = 101, 30, 5
z_all = torch.randn(n, REAL["Dz"], device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(seed))
tok, ep = model.sample_tokens(n, max_len=ml, z=z_all, seed=seed, row_offset=0)
want = model.vocabulary.decode_batch(tok.cpu().numpy(), ep.cpu().numpy())
got = model.sample_sharded(n, max_len=ml, seed=seed)
assert got == want, "sharded sampling differs from the single-process strings"
got2 = model.sample_sharded(n, max_len=ml)          # seed drawn on rank 0 and broadcast: the ranks still agree
both = [None, None]; dist.all_gather_object(both, got2); assert both[0] == both[1]

# generate loop under data parallelism: sharded sampling, scoring split over the ranks, one shuffle, DP fine-tune
from polygon_amd.vae.vae_generator import SmilesVaeMoleculeGenerator
class Obj:
    def __init__(self): self.n = 0
    def score_list(self, smiles):
        self.n += len(smiles)
        return [min(len(s), 30) / 30.0 for s in smiles]
obj = Obj()
out = os.path.join(%r, "gen%%d" %% rank); os.makedirs(out, exist_ok=True)
gen = SmilesVaeMoleculeGenerator(model, max_len=40, device=torch.device("cuda:0"), out_dir=out,
                                 canonicalizer=lambda s, include_stereocenters=True: [x for x in s if x])
gen.trainer.n_batch = 32
res = gen.optimise(obj, [], keep_top=128, n_epochs=2, mols_to_sample=256, optimize_n_epochs=2, optimize_batch_size=32,
                   pretrain_n_epochs=1, save_frequency=1)
mine = [(r.smiles, r.score) for r in res]
both = [None, None]; dist.all_gather_object(both, (mine, obj.n))
assert both[0][0] == both[1][0] and len(mine) > 0, "ranks disagree on the generation's results"
assert abs(both[0][1] - both[1][1]) <= 2            # each rank scored about half of every payload
assert os.path.exists(os.path.join(out, "GDM_2.pt")) == (rank == 0)
print("rank", rank, "generate ok", flush=True)

# a give-up on ONE rank: its gradients are garbage and have been summed into both ranks, so BOTH ranks must skip the
# update (weights stay equal across ranks) and BOTH must raise at their next check (nobody saves, nobody hangs)
import ctypes
from polygon_amd import _lib
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
assert L.pg_check_failure(_lib.stream_ptr()) == 0
before = eng.flat.clone()
# (this test runs with PG_NO_PERSIST=14 because its ranks share one GPU; rank 1 alone switches its persistent decoder
# kernels on for the failing step - they give up at their first unsatisfied wait, so residency does not matter)
old = L.pg_persist_spin_limit(ctypes.c_long(0)) if rank == 1 else None
if rank == 1:
    L.pg_persist_disable(0)
checks = L.pg_entry_checks(0)                 # as VAETrainer does under data parallelism: nobody raises mid-step
try:
    eng.train_step(sh, 0.4, opt.flat_grad, eps=eps, comm=comm); opt.step()
    torch.cuda.synchronize()
finally:
    L.pg_entry_checks(checks)
    if rank == 1:
        L.pg_persist_spin_limit(ctypes.c_long(old))
assert torch.equal(eng.flat, before), "rank %%d applied an update although a rank's recurrence gave up" %% rank
assert L.pg_check_failure(_lib.stream_ptr()) == -5, "rank %%d did not see the other rank's failure" %% rank
L.pg_persist_disable(0)
print("rank", rank, "shared failure ok", flush=True)
dist.barrier(); dist.destroy_process_group()
'''

NCCL1 = r'''
import os, sys, faulthandler
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from polygon_amd.parallel import init_distributed, GradComm
from polygon_amd.vae.vae_trainer import FusedClipAdam
from polygon_amd import _lib
from oracle import vae_oracle as O
from tests.helpers import REAL, build_model
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
model, _ = build_model(REAL, 21, "bf16", device="cuda:0", d_dropout=0.2)
model.train()
eng = model._engine
opt = FusedClipAdam(model, lr=3e-4)
strings = O.synthetic_smiles(509, seed=4) + ["C", "", "C" * 98]
gb = model.get_collate_fn()(list(strings))
eps = torch.randn(len(gb), REAL["Dz"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
assert _lib.lib().pg_persist_disable(-1) == 0          # the persistent kernels are the ones running
kl0, rec0 = eng.train_step(gb, 0.4, opt.flat_grad, eps=eps, p_drop=0.2, seed=99)
g0 = opt.flat_grad.clone()
comm = GradComm()
assert not comm.staged
for it in range(3):                                     # repeated: stream reuse, event re-recording
    kl, rec = eng.train_step(gb, 0.4, opt.flat_grad, eps=eps, p_drop=0.2, seed=99, comm=comm)
    torch.cuda.synchronize()
    err = ((opt.flat_grad - g0).abs().max() / g0.abs().max()).item()
    assert err <= 1e-4, err                             # identity all-reduce; fp32 atomics of the weight-gradient GEMMs
    assert abs(kl.item() - kl0.item()) <= 1e-6 * abs(kl0.item()) and abs(rec.item() - rec0.item()) <= 1e-6 * abs(rec0.item())
    opt.step()                                          # joins the side stream, runs clip + Adam
    _lib.check(_lib.lib().pg_check_failure(_lib.stream_ptr()), "after step")
    opt = FusedClipAdam(model, lr=0.0) if it == 0 else opt
    kl0, rec0 = eng.train_step(gb, 0.4, opt.flat_grad, eps=eps, p_drop=0.2, seed=99)
    g0 = opt.flat_grad.clone()
t = comm.sum_scalars(torch.tensor([2.5], device="cuda")); assert t.item() == 2.5
dist.barrier(); dist.destroy_process_group()
print("nccl world-1 ok", flush=True)
'''


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_shards_in_one_process_sum_to_whole_batch(dt):
    """The same equivalence without a second process: the two shards run one after the other through the PERSISTENT
    kernels with their data-parallel keys (global normalisers, dropout keyed by global row); their gradients and partial
    losses add up to the whole-batch step."""
    import numpy as np
    import torch
    from oracle import vae_oracle as O
    from polygon_amd.vae.vae_trainer import FusedClipAdam
    from tests.helpers import REAL, build_model
    model, _ = build_model(REAL, 21, dt, d_dropout=0.2)
    model.train()
    eng = model._engine
    opt = FusedClipAdam(model, lr=3e-4)
    strings = O.synthetic_smiles(300, seed=4) + ["C", ""]
    gb = model.get_collate_fn()(list(strings))
    B = len(gb)
    eps = torch.from_numpy(np.random.default_rng(1).standard_normal((B, REAL["Dz"])).astype(np.float32)).cuda()
    kl0, rec0 = eng.train_step(gb, 0.4, opt.flat_grad, eps=eps, p_drop=0.2, seed=99)
    g0 = opt.flat_grad.clone(); kl0 = kl0.item(); rec0 = rec0.item()
    world = 2
    g = torch.zeros_like(g0); kl = rec = 0.0
    for rank in range(world):
        sh = model.get_collate_fn(shard=(rank, world))(list(strings))
        same = gb.shard(rank, world)             # sharding after tokenising gives the same rows and the same pg_dp
        w = sh.tokens.shape[1]                  # a shard tokenised alone is only as wide as ITS longest row
        assert torch.equal(sh.tokens, same.tokens[:, :w]) and (same.tokens[:, w:] == model.pad).all()
        assert (sh.lengths == same.lengths).all()
        assert (sh.dp.kl_inv_batch, sh.dp.ce_count, sh.dp.b_mul, sh.dp.b_add) == \
            (same.dp.kl_inv_batch, same.dp.ce_count, same.dp.b_mul, same.dp.b_add)
        k, r = eng.train_step(sh, 0.4, opt.flat_grad, eps=eps[rank::world].contiguous(), p_drop=0.2, seed=99)
        g += opt.flat_grad; kl += k.item(); rec += r.item()
    err = ((g - g0).abs().max() / g0.abs().max()).item()
    assert err <= (1e-4 if dt == "fp32" else 2e-2), err
    assert abs(kl - kl0) <= 1e-4 * abs(kl0) + 1e-7
    assert abs(rec - rec0) <= (1e-4 if dt == "fp32" else 5e-3) * abs(rec0)


def test_nccl_world1_gradcomm_with_persistent_kernels(tmp_path):
    script = tmp_path / "nccl1.py"
    script.write_text(NCCL1 % ROOT)
    env = dict(os.environ)
    env.pop("PG_NO_PERSIST", None)
    r = subprocess.run([sys.executable, "-u", str(script)], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    assert "nccl world-1 ok" in r.stdout


def test_two_rank_step_equals_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, str(tmp_path), str(tmp_path)))
    # Two plain interpreter processes with the rendezvous variables set by hand: torch.distributed.run pulls in the whole
    # elastic-agent stack, whose first import on a fresh box (cold page cache) has taken more than two minutes.
    # The ranks share ONE GPU here, so they use the per-step decoder kernels (bits 2 | 4): a group-persistent launch needs
    # its whole grid resident, and two half-resident grids would wait for each other until their polling limits.  One
    # process per GPU - the deployment model - never meets this; the persistent kernels under data-parallel keys are
    # covered in-process above.
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", PG_NO_PERSIST="14", GLOO_SOCKET_IFNAME="lo")   # gloo must not resolve the hostname
        procs.append(subprocess.Popen([sys.executable, "-u", str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=130))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        # (round 1: a lowest-priority side stream in one rank was intermittently never scheduled beside the other rank's
        # queues - found through these stacks; the stream has default priority since)
        raise AssertionError("two ranks on one GPU stalled: %r" % ([p.communicate() for p in procs],))
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-3000:] + se[-5000:]
    assert sum(so.count("ok grad_rel_err") for so, _ in outs) == 4
    assert sum(so.count("trainer ok") for so, _ in outs) == 2
    assert sum(so.count("generate ok") for so, _ in outs) == 2
