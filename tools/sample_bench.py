import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polygon_amd.vae.vae_model import VAE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = VAE(compute_dtype=sys.argv[2] if len(sys.argv) > 2 else "bf16").cuda().eval()
z = torch.randn(n, 128, device="cuda")
m.sample_tokens(n, z=z, seed=1); torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(3): m.sample_tokens(n, z=z, seed=2 + r)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print("variant=%s n=%d: %.1f ms  %.0f mol/s  %.0f TF/s" % (os.environ.get("PG_FWD_VARIANT", "0"), n, dt * 1e3, n / dt, n / dt * 838e6 / 1e12))
