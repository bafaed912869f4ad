set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 100 --timeout-method thread 2>&1 | tail -3
timeout -k 10 120 python bench.py --steps 20 --warmup 3 --sample_n 0 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.3f  bwd_rec_us %.0f  fwd_rec_us %.0f' % (d['ms_per_step'], r['avg_launch_us'], r['fwd_recurrence']['avg_launch_us']))"
timeout -k 10 100 python tools/persist_stamps.py 1 0 20 2>&1 | grep -v amdgpu.ids | sed -n '1,3p;9p'
