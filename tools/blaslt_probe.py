"""Development aid: what the vendor GEMM library reaches on the weight-gradient shapes (torch.matmul -> hipBLASLt/rocBLAS)."""
import torch, time
def bench(M, N, R, iters=30):
    A = torch.randn(R, M, device="cuda").to(torch.bfloat16); B = torch.randn(R, N, device="cuda").to(torch.bfloat16)
    f = lambda: torch.matmul(A.t(), B)
    for _ in range(5): f()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); e.synchronize()
    ms = s.elapsed_time(e) / iters
    print("torch A^T B  M=%5d N=%5d R=%6d: %8.2f us  %7.1f TF/s" % (M, N, R, ms * 1e3, 2.0 * M * N * R / ms / 1e9), flush=True)
    try:
        f2 = lambda: torch.mm(A.t(), B, out_dtype=torch.float32)
        for _ in range(3): f2()
        s.record()
        for _ in range(iters): f2()
        e.record(); e.synchronize()
        ms = s.elapsed_time(e) / iters
        print("   fp32 out: %8.2f us  %7.1f TF/s" % (ms * 1e3, 2.0 * M * N * R / ms / 1e9), flush=True)
    except Exception as ex:
        print("   fp32 out unsupported:", type(ex).__name__, str(ex)[:100])
for pref in ("default",):
    bench(1536, 512, 50000); bench(768, 256, 50000); bench(64, 1536, 50000); bench(1536, 1024, 50000)
