# runs the quick backward check for library variants libpolygon_hip_<tag>.so (development aid)
set -e
for v in "$@"; do
  echo "== variant $v"
  cp polygon_amd/libpolygon_hip_$v.so polygon_amd/libpolygon_hip.so
  timeout -k 10 120 python tools/persist_err.py 1024 | tail -1 | cut -c1-90
  timeout -k 10 120 python bench.py --steps 30 --warmup 5 --sample_n 0 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.3f  tok/s %.0f  bwd_rec_us %.0f  fwd_rec_us %.0f' % (d['ms_per_step'], d['value'], r['avg_launch_us'], r['fwd_recurrence']['avg_launch_us']))"
done
