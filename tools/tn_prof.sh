#!/bin/bash
# kernel-level times of tools/tn_time.py (rocprofv3 kernel trace)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out
rm -rf $O/prof_tn
cd /tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/prof_tn -- python3 $R/tools/tn_time.py > $O/prof_tn.log 2>&1
cd $R
KT=$(find $O/prof_tn -name "*results.db" | head -1)
python3 tools/rocpd_summary.py stats "$KT" $O/tn.csv $O/tn.md "tn" > /dev/null
grep -E "gemm_tn|tn_reduce" $O/tn.md | cut -c1-160
grep "slab=" $O/prof_tn.log
rm -rf $O/prof_tn
