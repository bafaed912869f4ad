"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: identical global shuffles on every rank, length-balanced
sharding, global normalisers, two-bucket gradient all-reduce."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from polygon_amd.parallel import init_distributed, GradComm, global_batch_stats, world_info
from polygon_amd.vae.vae_model import VAE, TokenBatch
from polygon_amd.vae.vae_trainer import VAETrainer
rank, world, local = init_distributed(backend="gloo")
assert world == 2 and world_info() == (rank, 2)
# 1) two-bucket all-reduce == plain sum
g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
comm = GradComm()
comm.start_decoder_bucket(g, 300)
comm.finish(g, 300)
assert torch.equal(g, torch.arange(1000, dtype=torch.float32) * 3), "allreduce"
# 2) identical shuffled global batches on every rank; shards partition them and stay sorted
torch.manual_seed(1234 + rank)            # different local seeds: the loader must still agree (rank 0's seed is broadcast)
model = VAE()
tr = VAETrainer(model, n_batch=8)
data = ["C" * (1 + i %% 13) for i in range(100)]
loader = tr.get_dataloader(list(data), shuffle=True)
tot = 0
for batch in loader:
    lens = torch.from_numpy(batch.lengths.astype(np.int64))
    other = lens.clone()
    dist.broadcast(other, 0)
    assert torch.equal(lens, other), "global batches differ between ranks"
    assert len(batch) <= 16
    sh = batch.shard(rank, world)
    assert (np.diff(sh.lengths) <= 0).all()
    inv_b, cnt = global_batch_stats(batch.lengths)
    assert abs(inv_b - 1.0 / len(batch)) < 1e-12 and cnt == float(batch.lengths.sum() - len(batch))
    n_local = torch.tensor([len(sh)]); dist.all_reduce(n_local)
    assert int(n_local) == len(batch)
    tot += len(batch)
assert tot == 100
dist.barrier(); dist.destroy_process_group()
sys.stdout.write("rank %%d ok\n" %% rank); sys.stdout.flush()   # one write: the two ranks share the pipe
'''


def test_data_parallel_plumbing_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GLOO_SOCKET_IFNAME="lo")   # gloo must not resolve the hostname
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
