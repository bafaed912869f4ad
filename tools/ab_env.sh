#!/bin/bash
# A/B of environment switches on the training step (run through gpurun from the repo root):
#   bash tools/ab_env.sh "" "PG_TAIL_FORK_LATE=1" ...   -> ms_per_step of a short bench run under each setting
set -e
R=$PWD; O=$R/gpurun_out
: > $O/ab_env.txt
for e in "$@"; do
  line=$(env $e timeout -k 10 120 python3 $R/bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_e2e --no_proxy --no_roofline --sample_n 0 --no_configs0 2>/dev/null | tail -1)
  echo "[$e] $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" >> $O/ab_env.txt
done
cat $O/ab_env.txt
