"""Development aid: fixed cost of the TN GEMM (launch + prologue + split-K atomic epilogue) = its time with ONE row tile per block."""
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
    print("M=%5d N=%5d R=%6d: %8.2f us" % (M, N, R, s.elapsed_time(e) / iters * 1e3), flush=True)
for R in (16 * 32, 16 * 64, 16 * 128, 16 * 256, 16 * 512, 16 * 1024, 16 * 2048, 50000):
    bench(1536, 512, R)
