#!/bin/bash
# SQ counter passes (each its own run, <= 8 SQ counters) over a short bench run -> gpurun_out/r03_pmc_sq.json
set -e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out
rm -rf $O/prof_a $O/prof_b
cd /tmp
CMD="python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_e2e --no_proxy --sample_extra= --no_configs0"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/prof_a -- $CMD > $O/prof_a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES -d $O/prof_b -- $CMD > $O/prof_b.log 2>&1
cd $R
A=$(find $O/prof_a -name "*results.db" | head -1); B=$(find $O/prof_b -name "*results.db" | head -1)
python3 tools/rocpd_summary.py counters $O/r03_pmc_sq.json "rocprofv3 --pmc <8 SQ counters> (two separate passes) over python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline; per-launch averages" "$A" "$B"
rm -rf $O/prof_a $O/prof_b
python3 - <<'P'
import json
d=json.load(open('gpurun_out/r03_pmc_sq.json'))['bf16']
for k,v in d.items():
    print(k, {a:(round(b,3) if b<10 else int(b)) for a,b in v.items() if not a.startswith('launches_')})
P
