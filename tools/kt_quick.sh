# quick per-kernel times of the training step (development aid): rocprofv3 kernel trace over a short bench run
set +e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out
python3 tools/step_timeline.py "$KT" > $O/step_timeline.txt 2>&1 || true
rm -rf $O/prof_q
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_q -- python3 $R/bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_roofline --no_e2e --no_proxy --no_configs0 --sample_n 0 > $O/prof_q.log 2>&1
cd $R
KT=$(find $O/prof_q -name "*results.db" | head -1)
python3 tools/rocpd_summary.py stats "$KT" $O/q.csv $O/q.md "quick"
head -22 $O/q.md | cut -c1-200
python3 tools/step_timeline.py "$KT" > $O/step_timeline.txt 2>&1 || true
rm -rf $O/prof_q
