#!/bin/bash
# recurrence kernel time against batch size (run through gpurun): per-step fixed cost vs per-tile cost of the persistent kernels
set -e
for nb in 128 256 512 1024; do
  timeout -k 10 200 python bench.py --n_batch $nb --steps 20 --warmup 3 --no_cpu_baseline --no_e2e --no_proxy --sample_n 0 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('n_batch %s ms/step %.3f  bwd %.1f us  fwd %.1f us' % (sys.argv[1], d['ms_per_step'], r['avg_launch_us'], r['fwd_recurrence']['avg_launch_us']))" $nb
done
