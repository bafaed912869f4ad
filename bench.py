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


def step_kernel_microbench(dtype, B, H):
    """Isolated single-cell step kernels (gru_fwd2_kernel / gru_bwd2_kernel) at rows = B, a C-side loop of launches
    through pg_bench_gru timed with HIP events on the launch stream: the per-launch latency floor of the recurrence."""
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
    return {"rows": B, "flop_per_launch": flops,
            "bwd_cell": {"kernel": "gru_bwd2_kernel", "avg_launch_us": ms_bwd * 1e3, "tflops": flops / (ms_bwd * 1e-3) / 1e12},
            "fwd_cell": {"kernel": "gru_fwd2_kernel", "avg_launch_us": ms_fwd * 1e3, "tflops": flops / (ms_fwd * 1e-3) / 1e12}}


def profile_read(which):
    L = _lib.lib()
    ms, n, fl, seg = ctypes.c_double(), ctypes.c_long(), ctypes.c_double(), ctypes.c_int()
    _lib.check(L.pg_profile_read(which, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(seg)))
    return ms.value, n.value, fl.value, seg.value


def pmc_traffic(kernel, dtype):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 + WRITE_SIZE); None when not measured for dtype."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            t = json.load(f)
        for name, k in t.get(dtype, {}).items():     # keys are (possibly mangled) kernel names
            if kernel in name:
                return float(k["hbm_bytes_per_launch"])
        return None
    except (OSError, ValueError, KeyError):
        return None


def roofline_recurrence(dtype):
    """Dominant kernel (profiles/): the decoder's BPTT recurrence.  bf16 at the default sizes runs it as ONE group-persistent
    launch (dec_bwd_persist_kernel: every step of all layers, W_hh^T / W_ih_up^T stationary per workgroup); otherwise one
    gru_bwd2_diag_kernel launch per anti-diagonal of the (layer, t) grid.  Either way a cell is dh = dGH_{t+1}.W_hh
    (+ dGI_{l+1,t}.W_ih_{l+1}) on MFMA with the gate-gradient epilogue fused.  Measured LIVE over the timed region:
    pg_decoder_bwd brackets the recurrence (which launches nothing else) with a HIP event pair on the launch stream;
    avg launch = event time / launches; algorithmic flop = 2*3H*H*(rows_in + rows_out[layers below the top]) summed over
    the cells.  The forward recurrence is reported beside it the same way."""
    ms_b, n_b, fl_b, seg_b = profile_read(1)
    ms_f, n_f, fl_f, seg_f = profile_read(0)
    ach = fl_b / (ms_b * 1e-3) / 1e12
    k_b = "dec_bwd_persist_kernel" if n_b == seg_b else "gru_bwd2_diag_kernel"
    k_f = "dec_fwd_persist_kernel" if n_f == seg_f else "gru_fwd2_diag_kernel"
    what = {"dec_bwd_persist_kernel": " (whole decoder BPTT in one group-persistent launch: 3 layers x T steps, weights "
                                      "stationary in registers + LDS, activations all-gathered per step by LDS-DMA)",
            "gru_bwd2_diag_kernel": " (one BPTT anti-diagonal per launch)"}[k_b]
    return {"bound": "mfma", "kernel": k_b + what,
            "achieved": ach, "peak": PEAK[dtype], "unit": "TFLOP/s", "frac": ach / PEAK[dtype],
            "traffic": pmc_traffic(k_b, dtype),
            "avg_launch_us": ms_b / n_b * 1e3, "launches": n_b, "segments": seg_b, "flop_per_launch": fl_b / n_b,
            "fwd_recurrence": {"kernel": k_f, "avg_launch_us": ms_f / n_f * 1e3, "launches": n_f,
                               "flop_per_launch": fl_f / n_f, "achieved": fl_f / (ms_f * 1e-3) / 1e12,
                               "traffic": pmc_traffic(k_f, dtype)},
            "note": "the recurrence is a chain of T dependent steps of ~1-5 GFLOP each; inside the persistent launch a step is "
                    "bounded by the inter-workgroup hand-off (~3-5 us), the LDS-DMA intake of the all-gathered activations and "
                    "LDS fragment bandwidth (every wave reads the whole 16-row tile), not by MFMA throughput: DESIGN.md 3"}


def cpu_baseline(n_strings=1024):
    """The CPU oracle (numpy restatement of the reference's torch CPU path), one fwd+bwd+clip+Adam step, fp32, on ONE
    batch of the benchmark's size (bounded sample: ~5-15 s of CPU work on the GPU box's host cores)."""
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
    if rank == 0 and not args.no_roofline:   # two event records per recurrence per step, read after the timed region
        _lib.check(_lib.lib().pg_profile_enable(1))
        profile_read(0), profile_read(1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    roofline = None
    if rank == 0 and not args.no_roofline:
        _lib.check(_lib.lib().pg_profile_enable(0))
        roofline = roofline_recurrence(args.dtype)
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
                  "max_len": 100, "ms_per_batch": ds / reps * 1e3,
                  "kernel": "dec_sample_persist_kernel (the whole decode loop of a batch in one group-persistent launch: "
                            "3 GRU layers + vocab projection + softmax + inverse-CDF draw)" if args.dtype == "bf16" else
                            "gru_fwd2_kernel x3 + sample_head_kernel per step",
                  "tflops": mol_s * SAMPLE_FLOP_PER_MOL / 1e12,
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
        if roofline is not None:
            out["roofline"] = roofline
            out["roofline"]["single_cell_microbench"] = step_kernel_microbench(args.dtype, args.n_batch, 512)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
