"""Development aid: timeline of one steady-state training step from a rocprofv3 kernel-trace database.
usage: python tools/step_timeline.py <results.db> [step index from the end, default 2]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels")][0] if any(t.startswith("kernels") for t in tabs) else None
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
rows = list(db.execute("select name, start, end, queue_id, stream_id from %s order by start" % kd)) if "stream_id" in cols else \
       [r + (0,) for r in db.execute("select name, start, end, queue_id from %s order by start" % kd)]
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
i0, i1 = adam[-back - 1] + 1, adam[-back] + 1
t0 = rows[i0][1]
prev_end = t0
print("step of %d kernels, %.3f ms from first start to adam end" % (i1 - i0, (rows[i1 - 1][2] - t0) / 1e6))
for name, s, e, q, st in rows[i0:i1]:
    short = name.split("(")[0][-60:]
    print("%9.1f us  +%8.1f us  gap %7.1f  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, st, short))
    prev_end = max(prev_end, e)
