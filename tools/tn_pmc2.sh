#!/bin/bash
# SQ counters of the TN GEMM kernels over tools/tn_time.py (one pass, 8 SQ counters)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out
rm -rf $O/prof_tnp
cd /tmp
NOCS=1 timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $O/prof_tnp -- python3 $R/tools/tn_time.py > $O/prof_tnp.log 2>&1
cd $R
DB=$(find $O/prof_tnp -name "*results.db" | head -1)
python3 tools/rocpd_summary.py counters $O/tn_sq.json "tn" "$DB" > /dev/null 2>&1
python3 - <<'P'
import json
d=json.load(open('gpurun_out/tn_sq.json'))
d=d.get('bf16', d)
for k,v in d.items():
    if 'tn' in k: print(k[:40], {a:(round(b,3) if b<10 else int(b)) for a,b in v.items() if not a.startswith('launches_')})
P
rm -rf $O/prof_tnp
