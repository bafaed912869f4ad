#!/bin/bash
# timeline of one training step (run through gpurun from the repo root): kernel trace of a short bench run -> tools/step_timeline.py
set -e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out
rm -rf $O/prof_tl
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof_tl -- python3 $R/bench.py --steps 6 --warmup 3 --no_cpu_baseline --no_e2e --no_proxy --no_roofline --sample_n 0 --no_configs0 > $O/prof_tl.log 2>&1
cd $R
python3 tools/step_timeline.py $(find $O/prof_tl -name "*results.db" | head -1) 2 > $O/timeline.txt
rm -rf $O/prof_tl
