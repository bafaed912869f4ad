"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: identical global shuffles on every rank, sharding BEFORE
tokenising (each rank keeps rows rank::world of the length-sorted global batch), global normalisers carried by the shard,
a uniform rule for a tail smaller than the world, two-bucket gradient all-reduce, loss reduction."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from polygon_amd.parallel import init_distributed, GradComm, world_info
from polygon_amd.vae.vae_model import VAE, TokenBatch
from polygon_amd.vae.vae_trainer import VAETrainer
rank, world, local = init_distributed(backend="gloo")
assert world == 2 and world_info() == (rank, 2)
# 1) two-bucket all-reduce == plain sum; scalar reduction; broadcast
g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
comm = GradComm()
comm.start_decoder_bucket(g, 300)
comm.finish(g, 300)
assert torch.equal(g, torch.arange(1000, dtype=torch.float32) * 3), "allreduce"
v = torch.tensor([1.0 + rank, 10.0 * (rank + 1)])
assert comm.sum_scalars(v).tolist() == [3.0, 30.0]
b = torch.full((4,), float(rank + 5)); comm.broadcast(b, 0)
assert b.tolist() == [5.0] * 4
# 2) identical shuffled global batches on every rank; each rank tokenises only its rows; shards interleave to the sorted
#    global batch; the pg_dp a shard carries holds the GLOBAL normalisers
torch.manual_seed(1234 + rank)            # different local seeds: the loader must still agree (rank 0's seed is broadcast)
model = VAE()
tr = VAETrainer(model, n_batch=8)
for n_data, tail in ((100, 4), (97, 1)):
    data = ["C" * (1 + (7 * i) %% 13) + "N" * (i %% 3) for i in range(n_data)]
    loader = tr.get_dataloader(list(data), shuffle=True)
    tot = tot_local = 0
    seen = []
    for batch in loader:
        assert isinstance(batch, TokenBatch) and batch.dp is not None
        n_g = batch.global_rows
        both = [None, None]
        dist.all_gather_object(both, (batch.lengths.tolist(), batch.strings, n_g, batch.global_tokens))
        assert both[0][2] == both[1][2] and both[0][3] == both[1][3], "ranks disagree on the global batch"
        merged = [None] * n_g
        merged[0::2], merged[1::2] = both[0][0], both[1][0]
        assert all(a >= b for a, b in zip(merged, merged[1:])), "interleaved shards are not the sorted global batch"
        assert len(batch) == len(range(rank, n_g, 2)) and sum(merged) == batch.global_tokens
        assert batch.tokens.shape[0] == len(batch)        # only this rank's rows were tokenised
        assert abs(batch.dp.kl_inv_batch - 1.0 / n_g) < 1e-9 and batch.dp.ce_count == float(sum(merged) - n_g)
        assert (batch.dp.b_mul, batch.dp.b_add) == (2, rank)
        seen += both[0][1] + both[1][1]
        tot += n_g; tot_local += len(batch)
        last = (n_g, len(batch))
    assert tot == n_data and sorted(seen) == sorted(data)
    # the tail: 4 rows -> 2 each; 1 row -> rank 1 holds nothing, and BOTH ranks know (global_rows < world => both skip it)
    assert last == (tail, len(range(rank, tail, 2)))
# 3) eps of the global batch is the same on both ranks' generators (host stand-in for the device generator)
assert tr._shared_seed() == tr._shared_seed()
s = torch.tensor([tr._shared_seed()]); o = s.clone(); dist.broadcast(o, 0); assert torch.equal(s, o)
d = torch.tensor([int(torch.randint(0, 2 ** 62, (1,), generator=tr._seed_gen()))]); o = d.clone(); dist.broadcast(o, 0)
assert torch.equal(d, o), "dropout seeds differ between ranks"
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


def _launch_check(n):
    """`python bench.py --gpus n --launch_check` with no torchrun environment: the file starts n ranks by itself."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launch_check", "--backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_self_launch_two_ranks():
    out = _launch_check(2)
    assert out["launch_check"] and out["world"] == 2 and out["requested_gpus"] == 2 and out["self_launched"]
    assert sorted(r["rank"] for r in out["ranks"]) == [0, 1]
    assert sorted(r["local_rank"] for r in out["ranks"]) == [0, 1]
    assert all(r["world"] == 2 for r in out["ranks"])
    assert len({r["pid"] for r in out["ranks"]}) == 2          # two distinct child interpreters


def test_bench_self_launch_propagates_failure():
    """A rank that fails makes the launcher exit non-zero (here: an argument the children reject)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PG_BENCH_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch_check", "--backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
