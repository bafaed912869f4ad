#!/bin/bash
# A/B of the two-waves-per-SIMD forward kernel against the 4-wave one (same box, back to back)
for v in 0 1 0 1; do
  PG_FWD8_OFF=$v timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_e2e --no_proxy --sample_n 0 --no_configs0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
r = d['roofline']
print('PG_FWD8_OFF=$v step %.3f ms  bwd %.1f us  fwd %.1f us (%s)' % (d['ms_per_step'], r['avg_launch_us'], r['fwd_recurrence']['avg_launch_us'], r['fwd_recurrence']['kernel']))
" || exit 1
done
