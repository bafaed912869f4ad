"""Development aid: isolated time of the weight-gradient GEMM forms on the decoder's shape (1536 x 512 x ~50 k rows)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polygon_amd import _lib
M, N, R = 1536, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 50021
A = torch.randn(R, M, device="cuda").bfloat16(); B = torch.randn(R, N, device="cuda").bfloat16()
C = torch.zeros(M, N, device="cuda"); ca = torch.zeros(M, device="cuda")
SLAB = 776 * 128 * 128 * 4
slab = torch.empty(SLAB // 4, device="cuda")
L = _lib.lib()
def run(use_slab):
    _lib.check(L.pg_gemm_tn_ws(_lib.PG_BF16, _lib.ptr(A), ctypes.c_long(M), _lib.ptr(B), ctypes.c_long(N), M, N, R, _lib.ptr(C),
                               ctypes.c_long(N), ctypes.c_float(1.0), (_lib.ptr(ca) if not os.environ.get("NOCS") else None), None, _lib.ptr(slab) if use_slab else None,
                               ctypes.c_size_t(SLAB if use_slab else 0), _lib.stream_ptr()))
for use_slab in (0, 1):
    for _ in range(3): run(use_slab)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(use_slab)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("slab=%d PG_TN256=%s: %.1f us  %.0f TFLOP/s" % (use_slab, os.environ.get("PG_TN256", "1"), us, 2.0 * M * N * R / us / 1e6))
