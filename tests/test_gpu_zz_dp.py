"""Data-parallel equivalence on ONE GPU: two ranks (gloo, both on cuda:0) each run the fused step on their shard of a
global batch; the all-reduced flat gradient and the losses must equal the single-process step on the whole batch
(global normalisers, length-balanced sharding, dropout keyed by global row, two-bucket all-reduce)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, faulthandler
faulthandler.dump_traceback_later(50, exit=False)   # if a rank stalls, its stack goes to stderr before the parent gives up
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from polygon_amd.parallel import init_distributed, GradComm, global_batch_stats
from polygon_amd.vae.vae_model import _Dp
from polygon_amd.vae.vae_trainer import FusedClipAdam
from oracle import vae_oracle as O
from tests.helpers import REAL, build_model
rank, world, local = init_distributed(backend="gloo")
torch.cuda.set_device(0)
print("rank", rank, "init ok", flush=True)


class HostComm(GradComm):
    """Same two-bucket protocol as GradComm, reduced through host copies (gloo on CPU tensors): RCCL cannot run two
    ranks on one device, and the point here is the sharding / normalisation arithmetic."""
    def start_decoder_bucket(self, grads, dec_start):
        t = grads[dec_start:].cpu(); dist.all_reduce(t); grads[dec_start:].copy_(t)
    def finish(self, grads, dec_start):
        t = grads[:dec_start].cpu(); dist.all_reduce(t); grads[:dec_start].copy_(t)

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
    # sharded step
    inv_b, cnt = global_batch_stats(gb.lengths)
    dp = _Dp(kl_inv_batch=inv_b, ce_count=cnt, b_mul=world, b_add=rank)
    sh = gb.shard(rank, world)
    kl, rec = eng.train_step(sh, 0.4, opt.flat_grad, eps=eps[rank::world].contiguous(), dp=dp, p_drop=0.2, seed=99,
                             comm=HostComm())
    losses = torch.stack([kl, rec]).cpu()
    dist.all_reduce(losses)                      # per-rank partial losses (already globally normalised) sum to the global loss
    g = opt.flat_grad
    err = ((g - g0).abs().max() / g0.abs().max()).item()
    tol = 1e-4 if dt == "fp32" else 2e-2
    assert err <= tol, (dt, "grad", err)
    assert abs(losses[0].item() - kl0) <= 1e-4 * abs(kl0) + 1e-7, (dt, losses[0].item(), kl0)
    assert abs(losses[1].item() - rec0) <= (1e-4 if dt == "fp32" else 5e-3) * abs(rec0), (dt, losses[1].item(), rec0)
    print("rank", rank, dt, "ok grad_rel_err=%%.2e" %% err, flush=True)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_shards_in_one_process_sum_to_whole_batch(dt):
    """The same equivalence without a second process: the two shards run one after the other through the PERSISTENT
    kernels with their data-parallel keys (global normalisers, dropout keyed by global row); their gradients and partial
    losses add up to the whole-batch step."""
    import numpy as np
    import torch
    from oracle import vae_oracle as O
    from polygon_amd.parallel import global_batch_stats
    from polygon_amd.vae.vae_model import _Dp
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
    inv_b, cnt = global_batch_stats(gb.lengths)
    world = 2
    g = torch.zeros_like(g0); kl = rec = 0.0
    for rank in range(world):
        dp = _Dp(kl_inv_batch=inv_b, ce_count=cnt, b_mul=world, b_add=rank)
        k, r = eng.train_step(gb.shard(rank, world), 0.4, opt.flat_grad, eps=eps[rank::world].contiguous(), dp=dp, p_drop=0.2,
                              seed=99)
        g += opt.flat_grad; kl += k.item(); rec += r.item()
    err = ((g - g0).abs().max() / g0.abs().max()).item()
    assert err <= (1e-4 if dt == "fp32" else 2e-2), err
    assert abs(kl - kl0) <= 1e-4 * abs(kl0) + 1e-7
    assert abs(rec - rec0) <= (1e-4 if dt == "fp32" else 5e-3) * abs(rec0)


def test_two_rank_step_equals_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    # Two plain interpreter processes with the rendezvous variables set by hand: torch.distributed.run pulls in the whole
    # elastic-agent stack, whose first import on a fresh box (cold page cache) has taken more than two minutes.
    # The ranks share ONE GPU here, so they use the per-step decoder kernels (bits 2 | 4): a group-persistent launch needs
    # its whole grid resident, and two half-resident grids would wait for each other until their polling limits.  One
    # process per GPU - the deployment model - never meets this; the persistent kernels under data-parallel keys are
    # covered in-process below.
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", PG_NO_PERSIST="6", GLOO_SOCKET_IFNAME="lo")   # gloo must not resolve the hostname
        procs.append(subprocess.Popen([sys.executable, "-u", str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=75))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        # (round 1: a lowest-priority side stream in one rank was intermittently never scheduled beside the other rank's
        # queues - found through these stacks; the stream has default priority since)
        raise AssertionError("two ranks on one GPU stalled: %r" % ([p.communicate() for p in procs],))
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-3000:] + se[-5000:]
    assert sum(so.count("ok grad_rel_err") for so, _ in outs) == 4
