import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k, v in sorted(os.environ.items()):
    if k.split('_')[0] in ('HIP', 'AMD', 'GPU', 'HSA', 'ROCR', 'ROC', 'NCCL', 'RCCL', 'LD', 'OMP', 'MAX', 'PYTORCH', 'TORCH'):
        print(k, '=', v)
import torch
from polygon_amd import _lib
L = _lib.lib()
rows, H = 64, 512
nbytes = 80 * rows * H + 16 * H * H + (1 << 20)
ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()
def fn(n, which=0):
    _lib.check(L.pg_bench_gru(1, which, rows, H, _lib.ptr(ws), ctypes.c_size_t(nbytes), n, ctypes.c_void_p(st.cuda_stream)))
fn(50); torch.cuda.synchronize()
for n in (100, 1000):
    t0 = time.perf_counter(); fn(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("n=%d host enqueue %.2f us/launch, total to sync %.2f us/launch" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
x = torch.zeros(1, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000): x.add_(1)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("torch add_: host %.2f us/launch, to sync %.2f us" % ((t1 - t0) / 1000 * 1e6, (t2 - t0) / 1000 * 1e6))
