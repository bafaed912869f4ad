#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the bench command           -> gpurun_out/prof_kt
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes -> gpurun_out/prof_f, gpurun_out/prof_w
# and summarises them into gpurun_out/r03_*.{csv,md,json} (copy those into profiles/).
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
rm -rf $O/prof_kt $O/prof_f $O/prof_w
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_kt -- python3 $R/bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_e2e --no_proxy --sample_extra= --no_configs0 > $O/prof_kt.log 2>&1
tail -1 $O/prof_kt.log | cut -c1-300
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $O/prof_f -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_e2e --no_proxy --sample_extra= --no_configs0 > $O/prof_f.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $O/prof_w -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_e2e --no_proxy --sample_extra= --no_configs0 > $O/prof_w.log 2>&1
cd $R
KT=$(find $O/prof_kt -name "*results.db" | head -1); F=$(find $O/prof_f -name "*results.db" | head -1); W=$(find $O/prof_w -name "*results.db" | head -1)
echo "dbs: $KT $F $W"
python3 tools/rocpd_summary.py stats "$KT" $O/r03_bench_kernel_stats.csv $O/r03_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_e2e --no_proxy --sample_extra= --no_configs0 (bf16, n_batch 1024, includes the sampling and micro-bench legs)"
python3 tools/rocpd_summary.py traffic "$F" "$W" $O/r03_pmc_traffic.json bf16
head -12 $O/r03_bench_kernel_stats.md
# keep the merged-back output small: the databases stay on the box
rm -rf $O/prof_kt $O/prof_f $O/prof_w
