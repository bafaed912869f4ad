"""Data parallelism for VAE training: one process per GPU, gradients summed with RCCL over xGMI.

The reference has no multi-GPU code at all (polygon/run.py:611-612 is a comment).  Scheme: every rank
draws the SAME global batch of strings (same shuffle), sorts it on lengths only, keeps rows rank, rank+G, ... (its shard
stays sorted and length-balanced) and tokenises / uploads just those (VAE.get_collate_fn(shard=...)).  Losses are normalised by GLOBAL counts inside the kernels (pg_dp), so the
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
    """Two-bucket SUM all-reduce of the flat gradient buffer, decoder bucket overlapped with the encoder BPTT.

    The decoder bucket is reduced on a communication stream of its own: that stream waits for (a) everything the caller's
    stream has enqueued up to the end of pg_decoder_bwd and (b) the library's side stream, on which the decoder's large
    weight-gradient GEMMs run (pg_deferred_join) - so the GEMMs, the all-reduce and the encoder BPTT (caller's stream)
    overlap, and the caller's stream itself never waits for the GEMMs before the encoder BPTT.  ``finish`` reduces the
    encoder / embedding bucket and makes the caller's stream wait for both."""

    def __init__(self, group=None):
        self.group = group
        self.work = None
        self.stream = None
        self._flag = None
        # gloo (CPU rehearsals, two ranks sharing one GPU in the tests) moves device tensors through host copies; the
        # deployment backend is nccl (= RCCL on ROCm), which reduces in place on the device
        self.staged = dist.get_backend(group) != "nccl"

    def _join_deferred(self, stream):
        import ctypes
        from polygon_amd import _lib
        _lib.check(_lib.lib().pg_deferred_join(ctypes.c_void_p(stream.cuda_stream)), "pg_deferred_join")

    def _staged_sum(self, t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        t.copy_(h)

    def start_decoder_bucket(self, grads, dec_start):
        if not grads.is_cuda:
            self.work = dist.all_reduce(grads[dec_start:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return
        main = torch.cuda.current_stream(grads.device)
        if self.staged:
            self._join_deferred(main)
            self._staged_sum(grads[dec_start:])
            return
        if self.stream is None or self.stream.device != grads.device:
            self.stream = torch.cuda.Stream(device=grads.device)
        self.stream.wait_stream(main)
        self._join_deferred(self.stream)
        with torch.cuda.stream(self.stream):
            self.work = dist.all_reduce(grads[dec_start:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _share_failure(self, dev):
        """MAX over the ranks of the persistent kernels' give-up word (include/polygon_hip.h, pg_fail_flag_export): a
        rank whose recurrence gave up has already summed garbage into every rank's gradients, so every rank's Adam kernel
        must skip this update and every rank must raise at its next check.  Rides behind the encoder bucket."""
        import ctypes
        from polygon_amd import _lib
        L = _lib.lib()
        if self._flag is None or self._flag.device != dev:
            self._flag = torch.zeros(1, dtype=torch.int32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.pg_fail_flag_export(_lib.ptr(self._flag), st), "pg_fail_flag_export")
        if self.staged:
            h = self._flag.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            self._flag.copy_(h)
        else:
            dist.all_reduce(self._flag, op=dist.ReduceOp.MAX, group=self.group)
        _lib.check(L.pg_fail_flag_import(_lib.ptr(self._flag), st), "pg_fail_flag_import")

    def finish(self, grads, dec_start):
        if grads.is_cuda and self.staged:
            self._staged_sum(grads[:dec_start])
            self._share_failure(grads.device)
            return
        w2 = dist.all_reduce(grads[:dec_start], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.work is not None:
            self.work.wait()                 # the caller's (current) stream waits for the decoder bucket
            if grads.is_cuda:
                torch.cuda.current_stream(grads.device).wait_stream(self.stream)
            self.work = None
        w2.wait()
        if grads.is_cuda:
            self._share_failure(grads.device)

    def sum_scalars(self, t):
        """In-place SUM all-reduce of a small tensor of per-rank partial losses (the kernels normalise by GLOBAL counts,
        so the sum over ranks is the global-batch value)."""
        if t.is_cuda and self.staged:
            self._staged_sum(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def broadcast(self, t, src=0):
        if t.is_cuda and self.staged:
            h = t.cpu()
            dist.broadcast(h, src, group=self.group)
            t.copy_(h)
        else:
            dist.broadcast(t, src, group=self.group)
        return t


def global_batch_stats(lengths_global):
    """(1/B_global, number of CE targets) of a global batch, from its (host) lengths: every row contributes len-1
    non-pad targets (tokens produced by the tokeniser are never the pad id)."""
    B = int(lengths_global.shape[0])
    return 1.0 / B, float(int(lengths_global.sum()) - B)
