#!/bin/bash
# SQ counters of the TN (weight-gradient) GEMM alone (development aid) -> gpurun_out/tn_sq.txt
set -e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out
rm -rf $O/prof_a $O/prof_b
cd /tmp
timeout -s KILL 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/prof_a -- python3 $R/tools/tn_sweep.py one > $O/prof_a.log 2>&1
timeout -s KILL 100 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAVES -d $O/prof_b -- python3 $R/tools/tn_sweep.py one > $O/prof_b.log 2>&1
cd $R
python3 - <<P > $O/tn_sq.txt
import sqlite3, glob
for d in ("prof_a", "prof_b"):
    db = glob.glob("$O/%s/**/*results.db" % d, recursive=True)[0]
    c = sqlite3.connect(db)
    for r in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%gemm_tn%' group by counter_name"):
        print("%-36s %16.0f  (n=%d)" % r)
P
cat $O/tn_sq.txt
rm -rf $O/prof_a $O/prof_b
