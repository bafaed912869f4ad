#!/bin/bash
# knock-out timing of the 8-wave forward kernel (library built with -DPG_FWD8_KNOCKOUTS); results of KO != 0 are invalid
for ko in 0 16 2 4; do
  PG_FWD8_KO=$ko timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_e2e --no_proxy --sample_n 0 --no_configs0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
r = d['roofline']
print('KO=%2d fwd %.1f us' % ($ko, r['fwd_recurrence']['avg_launch_us']))
" || exit 1
done
