"""Micro-benchmark of the TN (weight-gradient) MFMA GEMM hook (development aid; knobs PG_TN_BLOCKS, PG_TN_BKR)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polygon_amd import _lib

def bench(M, N, R, iters=50):
    A = torch.randn(R, M, device="cuda").to(torch.bfloat16); B = torch.randn(R, N, device="cuda").to(torch.bfloat16)
    C = torch.zeros(M, N, device="cuda")
    L = _lib.lib(); st = torch.cuda.current_stream()
    def fn():
        _lib.check(L.pg_gemm_tn(_lib.PG_BF16, _lib.ptr(A), ctypes.c_long(M), _lib.ptr(B), ctypes.c_long(N), M, N, R, _lib.ptr(C),
                                ctypes.c_long(N), ctypes.c_float(1.0), ctypes.c_void_p(st.cuda_stream)))
    for _ in range(5): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); e.synchronize()
    ms = s.elapsed_time(e) / iters
    print("M=%5d N=%5d R=%6d: %8.2f us  %7.1f TF/s" % (M, N, R, ms * 1e3, 2.0 * M * N * R / ms / 1e9), flush=True)

print("PG_TN_BLOCKS=%s PG_TN_BKR=%s" % (os.environ.get("PG_TN_BLOCKS"), os.environ.get("PG_TN_BKR")))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    bench(1536, 512, 50000, iters=5)
else:
    bench(1536, 512, 50000)
    bench(768, 256, 50000)
    bench(64, 1536, 50000)
    bench(1536, 128, 1024)
