"""Micro-benchmark of the MFMA GEMM hook: per-iteration vs fixed cost (development aid)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polygon_amd import _lib

def bench(M, N, K, dt=torch.bfloat16, iters=300):
    code = _lib.PG_BF16 if dt == torch.bfloat16 else _lib.PG_F32
    A = torch.randn(M, K, device="cuda").to(dt); W = torch.randn(N, K, device="cuda").to(dt)
    C = torch.empty(M, N, device="cuda")
    L = _lib.lib(); st = torch.cuda.current_stream()
    def fn():
        _lib.check(L.pg_gemm_nt(code, _lib.ptr(A), ctypes.c_long(K), _lib.ptr(W), ctypes.c_long(K), M, N, K, None, _lib.ptr(C), None,
                                ctypes.c_long(N), 0, ctypes.c_float(1.0), ctypes.c_void_p(st.cuda_stream)))
    for _ in range(20): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); e.synchronize()
    ms = s.elapsed_time(e) / iters
    print("M=%6d N=%5d K=%5d %s: %9.2f us  %8.1f TF/s" % (M, N, K, str(dt)[6:], ms * 1e3, 2.0 * M * N * K / ms / 1e9))

for K in (64, 512, 2048, 8192):
    bench(1024, 1536, K)
for K in (64, 512, 2048):
    bench(128, 128, K)
bench(8192, 8192, 8192, iters=20)
bench(50000, 1536, 512, iters=50)
bench(4096, 4096, 4096, torch.float32, iters=20)
# empty-ish kernel launch overhead reference
x = torch.zeros(1, device="cuda")
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(1000): x.add_(1)
e.record(); e.synchronize(); print("torch tiny kernel: %.2f us/launch" % (s.elapsed_time(e)))
