import sys, os
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import importlib.util, torch
src = open("/root/repo/tools/gemm_sweep.py").read().split("for K in (64, 512, 2048, 8192):")[0]
exec(src)
bench(8192, 1536, 1024, iters=100)
bench(8192, 1536, 512, iters=100)
bench(8192, 8192, 8192, iters=10)
bench(50000, 1536, 512, iters=30)
