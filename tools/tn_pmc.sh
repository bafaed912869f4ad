#!/bin/bash
# L2-side counters of the TN GEMM alone (development aid)
set -e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out
rm -rf $O/prof_a $O/prof_b
cd /tmp
timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE -d $O/prof_a -- python3 $R/tools/tn_sweep.py one > $O/prof_a.log 2>&1
timeout -s KILL 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/prof_b -- python3 $R/tools/tn_sweep.py one > $O/prof_b.log 2>&1
cd $R
python3 - <<P > $O/tn_pmc.txt
import sqlite3, glob
for d in ("prof_a", "prof_b"):
    db = glob.glob("$O/%s/**/*results.db" % d, recursive=True)[0]
    c = sqlite3.connect(db)
    for r in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%gemm_tn%' group by counter_name"):
        print("%-36s %16.0f  (n=%d)" % r)
P
cat $O/tn_pmc.txt
rm -rf $O/prof_a $O/prof_b
