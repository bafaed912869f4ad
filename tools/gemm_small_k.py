"""Development aid: the two GEMMs around the cross entropy in isolation (the dout = dlogits . W_fc one has K = 64: an HBM-bound
write of the fp32 result)."""
import sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
src = open("/root/repo/tools/gemm_sweep.py").read().split("for K in (64, 512, 2048, 8192):")[0]
exec(src)
bench(22000, 512, 64, iters=100)
bench(22000, 64, 512, iters=100)
bench(22000, 512, 128, iters=100)
