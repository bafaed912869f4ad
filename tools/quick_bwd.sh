set -e
timeout -k 10 120 python tools/persist_err.py 1024 | tail -2
timeout -k 10 200 python -u -m pytest tests/test_gpu_persist.py -x -q --timeout 100 --timeout-method thread 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 120 python bench.py --steps 30 --warmup 5 --sample_n 0 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.3f  tok/s %.0f  bwd_rec_us %.0f  fwd_rec_us %.0f' % (d['ms_per_step'], d['value'], r['avg_launch_us'], r['fwd_recurrence']['avg_launch_us']))"
done
