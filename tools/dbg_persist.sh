set -e
for d in 0 8 10 14; do
  echo "== bwd PG_PERSIST_DBG=$d (2 no MFMA, 4 no epilogue I/O, 8 DMA re-reads tile 0)"
  PG_PERSIST_DBG=$d timeout -k 10 100 python tools/persist_stamps.py 1 1 60 2>&1 | grep -v amdgpu.ids | sed -n '3p;5p'
done
