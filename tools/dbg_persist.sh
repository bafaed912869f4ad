set -e
for d in 0 4 1 2 6 7; do
  echo "== fwd PG_PERSIST_DBG=$d (1 no DMA, 2 no MFMA, 4 stores dropped)"
  PG_NO_PERSIST=4 PG_PERSIST_DBG=$d timeout -k 10 100 python tools/persist_stamps.py 1 0 20 2>&1 | grep -v amdgpu.ids | sed -n '3p;5p'
done
