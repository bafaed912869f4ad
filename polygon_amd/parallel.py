"""Data parallelism for VAE training: one process per GPU, gradients summed with RCCL over xGMI.

The reference has no multi-GPU code at all (polygon/run.py:611-612 is a comment).  Scheme: every rank
collates the SAME global batch (same shuffle), which is sorted by length, and takes rows rank, rank+G, ... so its
shard stays sorted and length-balanced.  Losses are normalised by GLOBAL counts inside the kernels (pg_dp), so the
SUM all-reduce of the per-rank gradients equals the single-process gradient of the global batch.  The flat gradient
buffer is reduced in two buckets: the decoder bucket (~92 % of the bytes) as soon as the decoder BPTT has finished,
overlapping the encoder BPTT; the encoder/embedding bucket afterwards.  Clip + Adam then run redundantly per rank.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialises torch.distributed from the torchrun environment (RANK/WORLD_SIZE/LOCAL_RANK); no-op otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class GradComm:
    """Two-bucket SUM all-reduce of the flat gradient buffer, decoder bucket overlapped with the encoder BPTT."""

    def __init__(self, group=None):
        self.group = group
        self.work = None

    def start_decoder_bucket(self, grads, dec_start):
        self.work = dist.all_reduce(grads[dec_start:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, grads, dec_start):
        w2 = dist.all_reduce(grads[:dec_start], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.work is not None:
            self.work.wait()
            self.work = None
        w2.wait()


def global_batch_stats(lengths_global):
    """(1/B_global, number of CE targets) of a global batch, from its (host) lengths: every row contributes len-1
    non-pad targets (tokens produced by the tokeniser are never the pad id)."""
    B = int(lengths_global.shape[0])
    return 1.0 / B, float(int(lengths_global.sum()) - B)
