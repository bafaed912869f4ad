#!/bin/bash
# A/B of one environment switch on the training step: tools/ab_env.sh VAR=VALUE   (run through gpurun from the repo root)
set -e
run() { timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_e2e --no_proxy --sample_n 0 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%s ms/step %.3f  bwd %.1f us  fwd %.1f us  losses %s' % (sys.argv[1], d['ms_per_step'], r['avg_launch_us'], r['fwd_recurrence']['avg_launch_us'], d['last_losses']))" "$1"; }
run base
env "$1" bash -c "$(declare -f run); run '$1'"
run base2
