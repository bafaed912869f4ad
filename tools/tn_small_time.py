"""Development aid: isolated time of the small weight-gradient GEMMs of the step's tail (one-hot^T . dGI and friends)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polygon_amd import _lib
L = _lib.lib()
R = 48437
for (M, N, what) in ((64, 1536, "one-hot^T . dGI0 (decoder embedding table)"), (64, 768, "one-hot^T . dGI (encoder table)"),
                     (768, 256, "encoder W_hh gradient"), (1536, 128, "z projection (R = 1024)")):
    r = 1024 if "R = 1024" in what else R
    A = torch.randn(r, M, device="cuda").bfloat16(); B = torch.randn(r, N, device="cuda").bfloat16()
    C = torch.zeros(M, N, device="cuda")
    def run():
        _lib.check(L.pg_gemm_tn_ws(_lib.PG_BF16, _lib.ptr(A), ctypes.c_long(M), _lib.ptr(B), ctypes.c_long(N), M, N, r, _lib.ptr(C),
                                   ctypes.c_long(N), ctypes.c_float(1.0), None, None, None, ctypes.c_size_t(0), _lib.stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = (r * (M + N) * 2 + M * N * 4) / 1e6
    print("%-44s M %4d N %4d R %5d: %6.1f us  (%.0f MB -> %.2f TB/s, %.0f TFLOP/s)  PG_TN_BLOCKS=%s" % (what, M, N, r, us, mb, mb / us / 1e6 * 1e6 / 1e6, 2.0 * M * N * r / us / 1e6, os.environ.get("PG_TN_BLOCKS", "-")))
