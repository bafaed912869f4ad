"""Development aid: persistent sampler (pg_persist_sample.hip) against the per-step path on the same z / seed.
usage: python tools/sample_ab.py [n] [max_len]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from polygon_amd import _lib
from tests.helpers import REAL, build_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
max_len = int(sys.argv[2]) if len(sys.argv) > 2 else 100
model, _ = build_model(REAL, 31, "bf16", eos_bias=3.0)
model.eval()
L = _lib.lib()
z = torch.randn(n, REAL["Dz"], generator=torch.Generator().manual_seed(5)).cuda()

def run(mask, reps=3):
    old = L.pg_persist_disable(mask)
    try:
        tok, ep = model.sample_tokens(n, max_len=max_len, z=z, seed=77)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            tok, ep = model.sample_tokens(n, max_len=max_len, z=z, seed=77)
        torch.cuda.synchronize()
        return tok.cpu().numpy(), ep.cpu().numpy(), (time.perf_counter() - t0) / reps
    finally:
        L.pg_persist_disable(old)

t_ref, e_ref, dt_ref = run(8)
print("per-step  : %.2f ms  %.0f mol/s" % (dt_ref * 1e3, n / dt_ref), flush=True)
t_p, e_p, dt_p = run(0)
print("persistent: %.2f ms  %.0f mol/s" % (dt_p * 1e3, n / dt_p), flush=True)
same = (t_ref == t_p).all(1) & (e_ref == e_p)
print("identical rows: %d / %d (%.2f %%)" % (same.sum(), n, 100.0 * same.mean()))
first = np.where(~same)[0][:5]
for r in first:
    d = np.where(t_ref[r] != t_p[r])[0]
    print(" row %d first diff at col %s: ref %s pers %s  ends %d %d" % (r, d[:1], t_ref[r][d[:3]], t_p[r][d[:3]], e_ref[r], e_p[r]))
print("first-column agreement: %.4f" % (t_ref[:, 1] == t_p[:, 1]).mean())
