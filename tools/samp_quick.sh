#!/bin/bash
# sampler tests + the sampling legs of the bench (run through gpurun from the repo root)
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_persist.py tests/test_gpu_parity_pins.py -x -q --timeout 120 --timeout-method thread -k "sampler or sample or forced" > gpurun_out/t_samp.log 2>&1 || { tail -30 gpurun_out/t_samp.log; exit 1; }
tail -1 gpurun_out/t_samp.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_e2e --no_proxy --no_roofline --sample_extra "${1:-}" > gpurun_out/b_samp.log 2>&1
tail -1 gpurun_out/b_samp.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())['sample']
print('compact: %.0f mol/s  dev %.2f ms  steps/row %.1f | all steps: %.0f mol/s dev %.2f ms' % (d['value'], d['ms_per_batch_device_only'], d['mean_decode_steps_per_row'], d['all_steps']['value'], d['all_steps']['ms_per_batch_device_only']))
print(d.get('more_rows'))"
