"""Times the recurrent step kernels in a C-side launch loop (development aid)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polygon_amd import _lib
L = _lib.lib()
def run(which, rows, H, dt=_lib.PG_BF16, rep=300):
    nbytes = 80 * rows * H + 16 * H * H + (1 << 20)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    def fn(n):
        _lib.check(L.pg_bench_gru(dt, which, rows, H, _lib.ptr(ws), ctypes.c_size_t(nbytes), n, ctypes.c_void_p(st.cuda_stream)))
    fn(20); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); fn(rep); e.record(); e.synchronize()
    print("%s rows=%5d H=%4d: %7.2f us/launch" % ("fwd" if which == 0 else "bwd", rows, H, s.elapsed_time(e) / rep * 1e3))
for rows in (64, 1024):
    run(0, rows, 512); run(1, rows, 512)

