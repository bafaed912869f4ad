#!/usr/bin/env python3
"""Headline benchmark: VAE training tokens/sec (BASELINE.json configs[1]: n_batch 1024, d_dropout 0.2, bf16, one MI355X;
N>1: data-parallel over RCCL) plus sampled molecules/sec, on synthetic GuacaMol-shaped SMILES.

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]   (N>1: launched by torch.distributed.run)

A step = one pass of the training hot path over one batch already resident in HBM: encoder + decoder forward,
hand-written BPTT, gradient all-reduce (N>1), global-norm clip and Adam (vae_trainer.py:121-138).
``--scaling weak`` (default): n_batch rows PER GPU (global batch n_batch * N); ``strong``: the global batch stays n_batch
(north_star's configs[2] definition), each rank takes 1/N of it.  The line of the scaling not selected is measured in the
same run and reported as an extra object, so one driver run yields both.  Prints ONE JSON line on rank 0.

Extra legs (all reported inside the same line): ``e2e`` = VAETrainer.fit wall clock including DataLoader, length sort,
tokenising, pinned upload (the product's whole training path); ``sample`` = VAE.sample token matrices including the D2H
copy, at 8192 rows per GPU plus 65 536 and 1 048 576 rows at N = 1 (configs[3]); ``strong_scaling_proxy`` (N = 1) = the
per-rank step at the batch a rank would hold under strong scaling; ``cpu_baseline`` = the same hot path on the box's host
cores in stock torch.nn ops (what the reference runs) and in the numpy oracle.
"""
import argparse
import ctypes
import json
import os
import platform
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--n_batch", type=int, default=1024, help="batch per GPU (weak) / global batch (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--sample_n", type=int, default=8192, help="molecules per GPU in the sampling leg (0 = skip)")
    ap.add_argument("--sample_extra", default="65536,1048576", help="more row counts for the sampling leg at N = 1")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--no_e2e", action="store_true")
    ap.add_argument("--no_other_scaling", action="store_true")
    ap.add_argument("--no_proxy", action="store_true")
    ap.add_argument("--no_configs0", action="store_true", help="skip the fp32 configs[0] leg (n_batch 512)")
    ap.add_argument("--launch_check", action="store_true",
                    help="only start the ranks, form the process group, report rank/world per child and exit (no GPU work)")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"], help="default: nccl (= RCCL) when every rank has a GPU")
    return ap


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh interpreters of this file, one per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), relay rank 0's JSON line and fail if any rank
    fails.  Runs BEFORE torch or polygon_amd are imported: the parent never touches the GPU, and the children are plain
    subprocesses, never an exec of a process that has initialised HIP.  (The reference has no counterpart:
    polygon/run.py:611-612 is a comment.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PG_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if r == 0 else sys.stderr          # rank 0 prints the JSON line on the parent's stdout
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    pending = set(range(args.gpus))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
                for q in pending:
                    procs[q].terminate()             # exact children only
        time.sleep(0.05)
    return rc


if __name__ == "__main__":
    _args = build_parser().parse_args()
    if _args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(_args))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from polygon_amd import _lib  # noqa: E402
from polygon_amd.parallel import GradComm, init_distributed  # noqa: E402
from polygon_amd.utils.synthetic import synthetic_smiles  # noqa: E402
from polygon_amd.vae.vae_model import VAE  # noqa: E402
from polygon_amd.vae.vae_trainer import FusedClipAdam, VAETrainer  # noqa: E402

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
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/r0N_pmc_traffic.json, newest
    round first; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 + WRITE_SIZE); None when not measured."""
    for rnd in ("r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % rnd)) as f:
                t = json.load(f)
            for name, k in t.get(dtype, {}).items():     # keys are (possibly mangled) kernel names
                if kernel in name:
                    return float(k["hbm_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            continue
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
    traffic = pmc_traffic(k_b, dtype)
    sec = ms_b / n_b * 1e-3
    # the all-gather of the persistent kernel: every workgroup pulls its group's 3H-wide gate-gradient rows out of L2
    # (96 KB per 16-row tile below the top layer, 48 KB in the top layer: DESIGN.md 3); per-step kernels: not modelled
    l2_bytes = None
    if k_b == "dec_bwd_persist_kernel":
        operand_rows = fl_b / n_b / (2.0 * 3 * 512 * 512)          # gate-gradient rows consumed, summed over the cells
        l2_bytes = operand_rows * (3 * 512 * 2) * 8                # each is 3H bf16 wide and read by the layer's 8 unit slices
    return {"bound": "mfma", "kernel": k_b + what,
            "achieved": ach, "peak": PEAK[dtype], "unit": "TFLOP/s", "frac": ach / PEAK[dtype],
            "traffic": traffic,
            "frac_hbm": None if traffic is None else traffic / sec / 8.0e12,
            "l2_to_cu_bytes": l2_bytes, "frac_l2": None if l2_bytes is None else l2_bytes / sec / 34.5e12,
            "bound_in_practice": "neither MFMA nor HBM nor L2 bandwidth: the per-step hand-off between workgroups (~3 us x T steps) and "
                                 "the un-overlapped chain DMA landing -> MFMAs -> gate epilogue -> stores inside a tile; `frac` is "
                                 "the MFMA fraction the contract asks for, frac_hbm / frac_l2 the memory-side ones",
            "avg_launch_us": ms_b / n_b * 1e3, "launches": n_b, "segments": seg_b, "flop_per_launch": fl_b / n_b,
            "fwd_recurrence": {"kernel": k_f + (" (two waves per SIMD: dec_fwd8_persist_kernel)" if k_f == "dec_fwd_persist_kernel" and
                                                not os.environ.get("PG_FWD8_OFF") else ""),
                               "avg_launch_us": ms_f / n_f * 1e3, "launches": n_f,
                               "flop_per_launch": fl_f / n_f, "achieved": fl_f / (ms_f * 1e-3) / 1e12,
                               "traffic": pmc_traffic("dec_fwd8_persist_kernel", dtype) or pmc_traffic(k_f, dtype)},
            "note": "the recurrence is a chain of T dependent steps of ~1-5 GFLOP each; inside the persistent launch a step is "
                    "~3 us of inter-workgroup hand-off (poll, first tile's DMA, last tile's epilogue, store drain) plus ~3.3 us per "
                    "16-row BPTT tile whose 96 MFMAs per wave are 0.64 us of matrix pipe (16 cycles per v_mfma_f32_16x16x32_bf16, "
                    "tools/mfma_rate.hip) - the rest is DMA landing, LDS fragment latency, gate epilogue and stores on one wave "
                    "per SIMD: DESIGN.md 3, 7"}


# ---- CPU baseline (SURVEY.md 8d protocol) ------------------------------------------------------------------------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def _timed(fn, warm, timed, budget_s):
    """Median wall time of fn() over `timed` runs after `warm` warm-ups; stops early (>= 2 timed runs kept) once the
    leg has used `budget_s` seconds, so that the default bench stays within minutes on any host."""
    t_leg = time.perf_counter()
    for _ in range(warm):
        fn()
        if time.perf_counter() - t_leg > 0.4 * budget_s:
            break
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if len(ts) >= 2 and time.perf_counter() - t_leg > budget_s:
            break
    return statistics.median(ts), len(ts)


def cpu_baseline():
    """The training step (fwd + BPTT + clip 50 + Adam, d_dropout 0.2, fp32) and the 99-step sampler on the host cores of
    this box, in stock torch.nn ops (oracle/torch_cpu.py: what the reference's CPU path runs - nn.GRU over packed
    sequences, F.cross_entropy, torch.multinomial, optim.Adam) and in the numpy oracle (oracle/vae_oracle.py), on a
    BOUNDED sample of the benchmark's synthetic strings.  time.perf_counter wall clock, warm-up + timed runs, median."""
    from oracle import torch_cpu as TC
    from oracle import vae_oracle as O
    cores = os.cpu_count()
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    nthreads = int(os.environ.get("OMP_NUM_THREADS", 0)) or min(cores, 16)
    torch.set_num_threads(nthreads)
    P = O.init_params(seed=0)
    n_train = 512          # configs[0]'s batch
    seqs = O.collate(O.synthetic_smiles(n_train, seed=1))
    ntok = sum(len(s) for s in seqs)
    ts_ = TC.to_tensors(seqs)
    m = TC.StockVAE(P, d_dropout=0.2).train()
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    t_all, k_all = _timed(lambda: m.train_step(ts_, 0.5, opt), 3, 5, 30.0)
    # one thread, smaller sample
    seqs1 = O.collate(O.synthetic_smiles(64, seed=1))
    ntok1 = sum(len(s) for s in seqs1)
    ts1 = TC.to_tensors(seqs1)
    torch.set_num_threads(1)
    t_one, k_one = _timed(lambda: m.train_step(ts1, 0.5, opt), 1, 3, 10.0)
    torch.set_num_threads(nthreads)
    # numpy port, all BLAS threads
    eps = np.random.default_rng(0).standard_normal((n_train, 128)).astype(np.float32)
    st = O.adam_state(P)

    def np_step():
        enc, dec = O.vae_forward(P, seqs, eps)
        G, _ = O.vae_backward(P, enc, dec, g_kl=0.5, g_recon=1.0)
        G, _ = O.clip_grad_norm(G, 50.0)
        O.adam_step(P, G, st, lr=3e-4)
    t_np, k_np = _timed(np_step, 1, 3, 12.0)
    # sampler: 512 rows x 99 steps
    m.eval()
    z = torch.randn(512, 128)
    t_s, k_s = _timed(lambda: m.sample(z, max_len=100), 1, 3, 8.0)
    u = np.random.default_rng(1).random((99, 512), dtype=np.float32)
    P2 = O.init_params(seed=0)
    t_sn, k_sn = _timed(lambda: O.sample_tokens(P2, z.numpy(), u, max_len=100), 0, 2, 8.0)
    best_train = max(ntok / t_all, ntok / t_np)
    return {"value": best_train, "unit": "tokens/s", "cores": nthreads, "kind": "port",
            "cpu_model": _cpu_model(), "host_cores_visible": cores,
            "sample": "one train step (fwd+BPTT+clip+Adam, d_dropout 0.2, fp32) on %d synthetic strings = %d tokens; value = the "
                      "faster of the two restatements; median of the timed runs after warm-up" % (n_train, ntok),
            "train_torch_ops": {"tokens_per_s": ntok / t_all, "s_per_step": t_all, "threads": nthreads, "timed_runs": k_all,
                                "what": "stock torch.nn ops as the reference's CPU path (oracle/torch_cpu.py)"},
            "train_torch_ops_1thread": {"tokens_per_s": ntok1 / t_one, "s_per_step": t_one, "threads": 1, "timed_runs": k_one,
                                        "strings": 64, "tokens": ntok1},
            "train_numpy_port": {"tokens_per_s": ntok / t_np, "s_per_step": t_np, "threads": nthreads, "timed_runs": k_np,
                                 "what": "oracle/vae_oracle.py (numpy fp32, BLAS threads)"},
            "sample_torch_ops": {"molecules_per_s": 512 / t_s, "s_per_batch": t_s, "rows": 512, "max_len": 100, "timed_runs": k_s},
            "sample_numpy_port": {"molecules_per_s": 512 / t_sn, "s_per_batch": t_sn, "rows": 512, "max_len": 100,
                                  "timed_runs": k_sn}}


# ---- GPU legs --------------------------------------------------------------------------------------------------
def make_pool(model, strings, n_global, rank, world, n_pool, dev):
    """n_pool global batches of n_global strings; each rank tokenises and keeps ONLY its shard (sharding collate)."""
    collate = model.get_collate_fn(shard=(rank, world) if world > 1 else None)
    pool = []
    for i in range(n_pool):
        b = collate(list(strings[i * n_global:(i + 1) * n_global]))
        gen = torch.Generator(device=dev).manual_seed(1000 + i)          # the global batch's eps, same on every rank
        eps = torch.randn(b.global_rows, model.d_z, device=dev, generator=gen)[rank::world].contiguous()
        pool.append((b, eps))
    return pool


def time_steps(eng, opt, pool, comm, dropout, steps, warmup, barrier, profile=False):
    kl_w = 0.5

    def step(i):
        b, eps = pool[i % len(pool)]
        eng.train_step(b, kl_w, opt.flat_grad, eps=eps, p_drop=dropout, seed=1234 + i, comm=comm)
        opt.step()
    for i in range(warmup):
        step(i)
    barrier()
    if profile:   # two event records per recurrence per step, read after the timed region
        _lib.check(_lib.lib().pg_profile_enable(1))
        profile_read(0), profile_read(1)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if profile:
        _lib.check(_lib.lib().pg_profile_enable(0))
    tokens = sum(pool[(warmup + i) % len(pool)][0].global_tokens for i in range(steps))
    return dt, tokens


def max_over_ranks(dt, world, dev):
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def e2e_leg(args, dev, rank, world, barrier):
    """VAETrainer.fit over a synthetic slice: wall clock of the PRODUCT path including the DataLoader's shuffle, the
    length sort, tokenising, the pinned upload and the per-epoch loss read-back (vae_trainer.py:78-97,121-144)."""
    n_global = args.n_batch * world if args.scaling == "weak" else args.n_batch
    n_batches = 128 if world == 1 else 64
    strings = synthetic_smiles(n_batches * n_global, seed=7)
    ntok = sum(map(len, strings)) + 2 * len(strings)
    torch.manual_seed(0)
    model = VAE(d_dropout=args.dropout, compute_dtype=args.dtype).to(dev)
    kw = dict(n_batch=n_global // world, n_epoch=1, kl_start=0, kl_w_start=0.5, model_save=None, log_file=None)
    VAETrainer(model, **kw).fit(list(strings[:8 * n_global]))                      # warm-up (allocator, pinned pool)
    # host work per batch, in line (what the background thread hides): loader + sort + tokenise + pin + upload
    tr = VAETrainer(model, **kw)
    loader = tr.get_dataloader(list(strings[:16 * n_global]), shuffle=True)
    t0 = time.perf_counter()
    nb = sum(1 for _ in loader)
    torch.cuda.synchronize()
    host_ms = (time.perf_counter() - t0) / nb * 1e3
    out = {}
    for label, prefetch in (("prefetch2", 2), ("inline", 0)):
        tr = VAETrainer(model, prefetch=prefetch, **kw)
        barrier()
        t0 = time.perf_counter()
        tr.fit(list(strings))
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0, world, dev)
        out[label] = {"tokens_per_s": ntok / dt, "s_per_epoch": dt, "ms_per_batch": dt / n_batches * 1e3}
    # value = the product's default configuration (prefetch off: the background thread measured slower)
    return {"metric": "e2e_fit_tokens_per_sec", "value": out["inline"]["tokens_per_s"], "unit": "tokens/s",
            "strings": len(strings), "tokens": ntok, "batches": n_batches, "global_batch": n_global,
            "host_ms_per_batch_inline": host_ms, "with_background_collate": out["prefetch2"], "collate_in_line": out["inline"],
            "what": "VAETrainer.fit, 1 epoch, wall clock incl. DataLoader shuffle, length sort, tokenise, pinned upload, "
                    "fused step, epoch-end loss read-back; each rank tokenises only its shard"}


def configs0_leg(dev, barrier):
    """BASELINE.json configs[0] on the GPU at the reference's own precision: the MOSES-default VAE in fp32 (the reference's
    arithmetic, vae_model.py:66-98), n_batch 512, 10 000 synthetic strings, d_dropout 0.2 (the `polygon train` default,
    run.py:372-374): one `VAETrainer.fit` epoch (20 batches, the last of 272 strings) and the resident-batch step."""
    nb = 512
    strings = synthetic_smiles(10000, seed=20250310)
    ntok = sum(map(len, strings)) + 2 * len(strings)
    torch.manual_seed(0)
    model = VAE(d_dropout=0.2, compute_dtype="fp32").to(dev).train()
    opt = FusedClipAdam(model, lr=3e-4, clip_grad=50)
    eng = model._engine
    eng.ensure_shadow()
    pool = make_pool(model, strings, nb, 0, 1, 8, dev)
    d, t = time_steps(eng, opt, pool, None, 0.2, 20, 3, barrier)
    kw = dict(n_batch=nb, n_epoch=1, kl_start=0, kl_w_start=0.5, model_save=None, log_file=None)
    VAETrainer(model, **kw).fit(list(strings[:4 * nb]))          # warm-up
    barrier()
    t0 = time.perf_counter()
    VAETrainer(model, **kw).fit(list(strings))
    barrier()
    dt = time.perf_counter() - t0
    return {"workload": "configs[0]: MOSES-default SMILES VAE, fp32, n_batch 512, d_dropout 0.2, 10 000 synthetic strings",
            "dtype": "fp32", "n_batch": nb,
            "resident_step": {"ms_per_step": d / 20 * 1e3, "tokens_per_s": t / d,
                              "frac_of_fp32_matrix_peak": t / d * TRAIN_FLOP_PER_TOKEN / 1e12 / PEAK["fp32"]},
            "fit_epoch": {"strings": len(strings), "tokens": ntok, "s_per_epoch": dt, "tokens_per_s": ntok / dt},
            "kernels": "per-step gru_fwd2/gru_bwd2 diagonals on v_mfma_f32_32x32x2_f32 (the persistent recurrences are bf16-only)"}


def sample_leg(model, n, rank, world, dev, barrier, reps=3, all_steps=False):
    """n rows per GPU, max_len 100: token matrix + end positions INCLUDING the D2H copy (SURVEY.md 8d metric 2).
    Rows that have drawn their eos leave the decode (pg_persist_sample.hip, build_live_list); ``all_steps`` switches that
    off (every row runs all 99 steps, as in round 1).  The flop rate counts the steps rows really ran."""
    z = torch.randn(n, model.d_z, device=dev)
    if all_steps:
        os.environ["PG_SAMPLE_NO_COMPACT"] = "1"
    try:
        def once(seed):
            tok, ep = model.sample_tokens(n, max_len=100, z=z, seed=seed, row_offset=rank * n)
            return tok.cpu(), ep.cpu()
        once(1)
        barrier()
        t1 = time.perf_counter()
        for r in range(reps):
            tok, ep = once(2 + r)
        barrier()
        ds = max_over_ranks(time.perf_counter() - t1, world, dev)
        # device-only time of the same call, for the split
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        model.sample_tokens(n, max_len=100, z=z, seed=9, row_offset=rank * n)
        torch.cuda.synchronize()
        dev_ms = (time.perf_counter() - t2) * 1e3
    finally:
        os.environ.pop("PG_SAMPLE_NO_COMPACT", None)
    mol_s = reps * n * world / ds
    # decode steps a row runs: up to and including the one that draws its eos (end position - 1), 99 without one
    steps_run = 99.0 if all_steps else float((ep.double() - 1).clamp(max=99).mean())
    return {"value": mol_s, "unit": "molecules/s", "n_per_gpu": n, "max_len": 100, "ms_per_batch": ds / reps * 1e3,
            "ms_per_batch_device_only": dev_ms, "includes_d2h": True, "mean_decode_steps_per_row": steps_run,
            "tflops": mol_s * SAMPLE_FLOP_PER_MOL * steps_run / 99.0 / 1e12}


def launch_check(args):
    """Every rank joins the process group and reports (rank, world, local rank, pid); rank 0 prints them as ONE JSON line."""
    import torch.distributed as dist
    if os.environ.get("PG_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):      # test hook: this rank fails before the rendezvous
        sys.exit(7)
    want_nccl = args.backend == "nccl" or (args.backend is None and torch.cuda.is_available()
                                           and torch.cuda.device_count() >= int(os.environ.get("WORLD_SIZE", "1")))
    rank, world, local = init_distributed("nccl" if want_nccl else "gloo")
    me = [rank, world, local, os.getpid()]
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, me)
    else:
        got = [me]
    if rank == 0:
        print(json.dumps({"launch_check": True, "requested_gpus": args.gpus, "world": world,
                          "backend": dist.get_backend() if world > 1 else None,
                          "rccl_ranks": world if world > 1 and dist.get_backend() == "nccl" else 0,
                          "self_launched": os.environ.get("PG_BENCH_SELF_LAUNCHED") == "1",
                          "ranks": [{"rank": r, "world": w, "local_rank": lr, "pid": p} for r, w, lr, p in got]}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if world == args.gpus else 3


def main():
    args = build_parser().parse_args()
    if args.launch_check:
        sys.exit(launch_check(args))

    rank, world, local = init_distributed(args.backend)
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but the process group has %d ranks" % (args.gpus, world), file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(0)
    model = VAE(d_dropout=args.dropout, compute_dtype=args.dtype).to(dev).train()
    opt = FusedClipAdam(model, lr=3e-4, clip_grad=50)
    eng = model._engine
    eng.ensure_shadow()
    comm = GradComm() if world > 1 else None
    if comm is not None:
        comm.broadcast(eng.flat, 0)
        eng.invalidate()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    n_pool = 8

    def run_scaling(mode, profile):
        n_global = args.n_batch * world if mode == "weak" else args.n_batch
        strings = synthetic_smiles(n_pool * n_global, seed=20250310)
        pool = make_pool(model, strings, n_global, rank, world, n_pool, dev)
        dt, tokens = time_steps(eng, opt, pool, comm, args.dropout, args.steps, args.warmup, barrier, profile)
        return max_over_ranks(dt, world, dev), tokens, n_global

    profile = rank == 0 and not args.no_roofline
    dt, tokens, n_global = run_scaling(args.scaling, profile)
    roofline = roofline_recurrence(args.dtype) if profile else None
    tok_per_s = tokens / dt
    kl, recon = eng._small[2].tolist()
    other = None
    if world > 1 and not args.no_other_scaling:
        mode = "strong" if args.scaling == "weak" else "weak"
        dt2, tokens2, ng2 = run_scaling(mode, False)
        other = {"scaling": mode, "value": tokens2 / dt2, "unit": "tokens/s", "ms_per_step": dt2 / args.steps * 1e3,
                 "global_batch": ng2, "rows_per_gpu": ng2 // world}
    proxy = None
    if world == 1 and not args.no_proxy:   # the step a rank would run under strong scaling of the 1024-row global batch (no all-reduce)
        proxy = {}
        for nb in (512, 256, 128):
            strings = synthetic_smiles(n_pool * nb, seed=20250310)
            pool = make_pool(model, strings, nb, 0, 1, n_pool, dev)
            d, t = time_steps(eng, opt, pool, None, args.dropout, 20, 3, barrier)
            proxy["n_batch_%d" % nb] = {"ms_per_step": d / 20 * 1e3, "tokens_per_s": t / d}
    large = None
    if world == 1 and not args.no_proxy:   # above the persistent kernels' 1024-row limit: interleaved micro-batches (vae_model.py)
        nb = 2048
        strings = synthetic_smiles(4 * nb, seed=20250311)
        pool = make_pool(model, strings, nb, 0, 1, 4, dev)
        d, t = time_steps(eng, opt, pool, None, args.dropout, 10, 2, barrier)
        large = {"n_batch_%d" % nb: {"ms_per_step": d / 10 * 1e3, "tokens_per_s": t / d, "micro_batches": eng.micro_batches(nb)}}

    e2e = None
    if not args.no_e2e:
        e2e = e2e_leg(args, dev, rank, world, barrier)
    configs0 = None
    if world == 1 and not args.no_configs0:
        configs0 = configs0_leg(dev, barrier)

    # ---- sampling leg (configs[3]): per-GPU shard, no collective during decode --------------------------------
    sample = None
    if args.sample_n > 0:
        model.eval()
        sample = sample_leg(model, args.sample_n, rank, world, dev, barrier)
        full = sample_leg(model, args.sample_n, rank, world, dev, barrier, all_steps=True)
        sample.update({"metric": "sampled_molecules_per_sec",
                       "rows_leave_at_eos": "rows that have drawn eos are dropped from the decode (same tokens); the model is the "
                                            "benchmark's random-init one after its few training steps, whose rows end by "
                                            "chance (mean_decode_steps_per_row); all_steps = the same call with every row "
                                            "decoded to max_len, the round-1 behaviour",
                       "all_steps": {k: full[k] for k in ("value", "ms_per_batch", "ms_per_batch_device_only", "tflops")},
                       "kernel": "dec_sample_persist_kernel (the whole decode loop of up to 16384 rows in one group-persistent "
                                 "launch: 3 GRU layers + vocab projection + softmax + inverse-CDF draw)" if args.dtype == "bf16"
                                 else "gru_fwd2_kernel x3 + sample_head_kernel per step",
                       "frac_of_mfma_peak": sample["tflops"] / (PEAK[args.dtype] * world),
                       "frac_of_mfma_peak_all_steps": full["tflops"] / (PEAK[args.dtype] * world)})
        if world == 1 and args.sample_extra:
            sample["more_rows"] = {}
            for n in [int(x) for x in args.sample_extra.split(",") if x]:
                s = sample_leg(model, n, 0, 1, dev, barrier, reps=2 if n > 100000 else 3)
                sample["more_rows"]["n_%d" % n] = {k: s[k] for k in ("value", "ms_per_batch", "ms_per_batch_device_only",
                                                                      "mean_decode_steps_per_row")}

    if rank == 0:
        out = {
            "metric": "vae_train_tokens_per_sec", "value": tok_per_s, "unit": "tokens/s", "n_gpus": world,
            "rccl_ranks": world if world > 1 and torch.distributed.get_backend() == "nccl" else 0,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: MOSES-default SMILES VAE (GRU enc 256, 3xGRU dec 512, d_z 128, vocab 51) "
                                   "pretraining step, %s, d_dropout %.2f, %s, synthetic GuacaMol-shaped strings (mean 49 "
                                   "tokens), fwd+BPTT+clip+Adam" % (
                                       "n_batch %d per GPU" % args.n_batch if args.scaling == "weak" else
                                       "global batch %d split over the GPUs" % args.n_batch, args.dropout, args.dtype),
                       "global_batch": n_global, "parallelism": "dp%d" % world},
            "train_tflops": tok_per_s * TRAIN_FLOP_PER_TOKEN / 1e12,
            "train_frac_of_mfma_peak": tok_per_s * TRAIN_FLOP_PER_TOKEN / 1e12 / (PEAK[args.dtype] * world),
            "last_losses": {"kl": kl, "recon": recon},
            "sample": sample,
        }
        if other is not None:
            out["other_scaling"] = other
        if proxy is not None:
            out["strong_scaling_proxy"] = proxy
        if large is not None:
            out["large_batch"] = large
        if e2e is not None:
            out["e2e"] = e2e
        if configs0 is not None:
            out["configs0"] = configs0
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
