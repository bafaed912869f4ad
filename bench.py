#!/usr/bin/env python3
"""Headline benchmark: VAE training tokens/sec (BASELINE.json configs[1]: n_batch 1024, d_dropout 0.2, bf16, one MI355X;
N>1: the same per-GPU batch data-parallel over RCCL) plus sampled molecules/sec, on synthetic GuacaMol-shaped SMILES.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the training hot path over one batch already resident in HBM: encoder + decoder forward,
hand-written BPTT, gradient all-reduce (N>1), global-norm clip and Adam (vae_trainer.py:121-138).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from polygon_amd import _lib  # noqa: E402
from polygon_amd.parallel import GradComm, global_batch_stats, init_distributed  # noqa: E402
from polygon_amd.utils.synthetic import synthetic_smiles  # noqa: E402
from polygon_amd.vae.vae_model import VAE, TokenBatch, _Dp  # noqa: E402
from polygon_amd.vae.vae_trainer import FusedClipAdam  # noqa: E402

TRAIN_FLOP_PER_TOKEN = 26.8e6       # 4 468 992 MAC fwd x 2 flop x 3 (fwd + bwd)
SAMPLE_FLOP_PER_MOL = 838e6         # 99 steps x 4 233 216 MAC x 2
PEAK = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA TFLOP/s (MI355X_MICROARCH.md)


def hip_events_ms(fn, stream, iters):
    """Average duration of fn() in ms, measured with HIP events recorded on `stream`."""
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        fn()
        start.record(stream)
        for _ in range(iters):
            fn()
        end.record(stream)
    end.synchronize()
    return start.elapsed_time(end) / iters


def roofline_gate_gemm(dtype, B, H):
    """Dominant kernel (rocprof: profiles/): the BPTT step kernel gru_bwd2_kernel = gate-gradient GEMM
    [B,3H]x[3H,H] on MFMA + fused gate-gradient epilogue.  Timed live with HIP events on the stream it is launched on
    (a C-side loop of launches through pg_bench_gru, rows = B).  Algorithmic flop per launch = 2*B*3H*H."""
    code = _lib.PG_BF16 if dtype == "bf16" else _lib.PG_F32
    L = _lib.lib()
    nbytes = 80 * B * H + 16 * H * H + (1 << 20)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    reps = 200

    def run(which):
        def fn():
            _lib.check(L.pg_bench_gru(code, which, B, H, _lib.ptr(ws), ctypes.c_size_t(nbytes), reps,
                                      ctypes.c_void_p(st.cuda_stream)))
        return hip_events_ms(fn, st, 3) / reps
    ms_bwd, ms_fwd = run(1), run(0)
    flops = 2.0 * B * H * 3 * H
    ach = flops / (ms_bwd * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "gru_bwd2_kernel (BPTT step: dGH[%d,%d] x W_hh[%d,%d] + gate-gradient epilogue)" % (B, 3 * H, 3 * H, H),
            "achieved": ach, "peak": PEAK[dtype], "unit": "TFLOP/s", "frac": ach / PEAK[dtype], "traffic": None,
            "avg_launch_us": ms_bwd * 1e3, "flop_per_launch": flops,
            "fwd_step_kernel": {"kernel": "gru_fwd2_kernel", "avg_launch_us": ms_fwd * 1e3,
                                "achieved": flops / (ms_fwd * 1e-3) / 1e12},
            "note": "step kernels are latency-bound (launch floor ~2.6 us + operand fetch + short K loop), see DESIGN.md 3"}


def cpu_baseline(n_strings=192):
    """The CPU oracle (numpy restatement of the reference's torch CPU path), one fwd+bwd+clip+Adam step, fp32."""
    from oracle import vae_oracle as O
    try:
        import threadpoolctl
        nthreads = max(i["num_threads"] for i in threadpoolctl.threadpool_info() or [{"num_threads": 1}])
    except Exception:
        nthreads = os.cpu_count()
    P = O.init_params(seed=0)
    strings = O.synthetic_smiles(n_strings, seed=1)
    seqs = O.collate(strings)
    ntok = sum(len(s) for s in seqs)
    eps = np.random.default_rng(0).standard_normal((n_strings, 128)).astype(np.float32)
    st = O.adam_state(P)
    t0 = time.perf_counter()
    enc, dec = O.vae_forward(P, seqs, eps)
    G, _ = O.vae_backward(P, enc, dec, g_kl=1.0, g_recon=1.0)
    G, _ = O.clip_grad_norm(G, 50.0)
    O.adam_step(P, G, st, lr=3e-4)
    dt = time.perf_counter() - t0
    return {"value": ntok / dt, "unit": "tokens/s", "cores": int(min(nthreads, os.cpu_count())), "kind": "port",
            "sample": "1 train step (fwd+BPTT+clip+Adam) of oracle/vae_oracle.py (numpy fp32) on %d synthetic strings, "
                      "%d tokens, %.1f s wall" % (n_strings, ntok, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--n_batch", type=int, default=1024, help="per-GPU batch (weak scaling)")
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--sample_n", type=int, default=8192, help="molecules per GPU in the sampling leg (0 = skip)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    args = ap.parse_args()

    rank, world, local = init_distributed()
    if world != args.gpus and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(0)
    model = VAE(d_dropout=args.dropout, compute_dtype=args.dtype).to(dev).train()
    opt = FusedClipAdam(model, lr=3e-4, clip_grad=50)
    eng = model._engine
    eng.ensure_shadow()

    # synthetic global batches, identical on every rank; each rank keeps its shard resident in HBM
    n_pool = 8
    Bg = args.n_batch * world
    strings = synthetic_smiles(n_pool * Bg, seed=20250310)
    collate = model.get_collate_fn()
    pool = []
    for i in range(n_pool):
        gb = collate(strings[i * Bg:(i + 1) * Bg])
        dp = None
        tokens_global = int(gb.lengths.sum())
        if world > 1:
            inv_b, cnt = global_batch_stats(gb.lengths)
            dp = _Dp(kl_inv_batch=inv_b, ce_count=cnt, b_mul=world, b_add=rank)
            gb = gb.shard(rank, world)
        eps = torch.randn(len(gb), model.d_z, device=dev)
        pool.append((gb, dp, eps, tokens_global))
    comm = GradComm() if world > 1 else None
    kl_w = 0.5

    def step(i):
        b, dp, eps, _ = pool[i % n_pool]
        eng.train_step(b, kl_w, opt.flat_grad, eps=eps, dp=dp, p_drop=args.dropout, seed=1234 + i, comm=comm)
        opt.step()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    tokens = sum(pool[(args.warmup + i) % n_pool][3] for i in range(args.steps))
    tok_per_s = tokens / dt
    kl, recon = eng._small[2].tolist()

    # ---- sampling leg (cfg4): per-GPU shard, no collective ------------------------------------------------
    sample = None
    if args.sample_n > 0:
        model.eval()
        z = torch.randn(args.sample_n, model.d_z, device=dev)
        model.sample_tokens(args.sample_n, max_len=100, z=z, seed=1, row_offset=rank * args.sample_n)
        barrier()
        t1 = time.perf_counter()
        reps = 3
        for r in range(reps):
            tok, ep = model.sample_tokens(args.sample_n, max_len=100, z=z, seed=2 + r, row_offset=rank * args.sample_n)
        barrier()
        ds = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([ds], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            ds = float(t.item())
        mol_s = reps * args.sample_n * world / ds
        sample = {"metric": "sampled_molecules_per_sec", "value": mol_s, "unit": "molecules/s", "n_per_gpu": args.sample_n,
                  "max_len": 100, "tflops": mol_s * SAMPLE_FLOP_PER_MOL / 1e12,
                  "frac_of_mfma_peak": mol_s * SAMPLE_FLOP_PER_MOL / 1e12 / (PEAK[args.dtype] * world)}

    if rank == 0:
        out = {
            "metric": "vae_train_tokens_per_sec", "value": tok_per_s, "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: MOSES-default SMILES VAE (GRU enc 256, 3xGRU dec 512, d_z 128, vocab 51) "
                                   "pretraining step, n_batch %d per GPU, d_dropout %.2f, %s, synthetic GuacaMol-shaped "
                                   "strings (mean 49 tokens), fwd+BPTT+clip+Adam" % (args.n_batch, args.dropout, args.dtype),
                       "global_batch": Bg, "parallelism": "dp%d" % world},
            "train_tflops": tok_per_s * TRAIN_FLOP_PER_TOKEN / 1e12,
            "train_frac_of_mfma_peak": tok_per_s * TRAIN_FLOP_PER_TOKEN / 1e12 / (PEAK[args.dtype] * world),
            "last_losses": {"kl": kl, "recon": recon},
            "sample": sample,
        }
        if not args.no_roofline:
            out["roofline"] = roofline_gate_gemm(args.dtype, args.n_batch, 512)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
