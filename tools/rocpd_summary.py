#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite output) runs into the files kept under profiles/.

  python tools/rocpd_summary.py stats  <kernel-trace results.db> <out.csv> <out.md> [title]
  python tools/rocpd_summary.py traffic <FETCH_SIZE results.db> <WRITE_SIZE results.db> <out.json> <dtype>

`traffic` applies the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-byte
read requests at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Both counters are in KiB
and sit on the L2's memory side, i.e. Infinity-Cache hits are included."""
import json
import sqlite3
import sys


def stats(db, out_csv, out_md, title):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out_csv, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
    with open(out_md, "w") as f:
        f.write("# " + title + "\n\n`avg w/o max` drops the slowest launch (the cold first launch of a persistent kernel loads its code object and "
                "weights).\n\n| kernel | calls | total ms | avg us | avg w/o max us | min us | max us | pct |\n|---|---:|---:|---:|---:|---:|---:|---:|\n")
        for r in rows[:28]:
            a2 = (r[2] - r[5]) / (r[1] - 1) / 1e3 if r[1] > 1 else r[2] / 1e3
            f.write("| `%s` | %d | %.2f | %.2f | %.2f | %.2f | %.2f | %.1f |\n" % (r[0][:100], r[1], r[2] / 1e6, r[3] / 1e3, a2,
                                                                                r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))


def traffic(db_f, db_w, out_json, dtype):
    def per_kernel(db, name):
        c = sqlite3.connect(db)
        return {r[0]: (r[1], r[2]) for r in c.execute(
            "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (name,))}
    F, W = per_kernel(db_f, "FETCH_SIZE"), per_kernel(db_w, "WRITE_SIZE")
    out = {}
    for k in sorted(set(F) | set(W)):
        short = k
        for tag in ("dec_sample_persist_kernel", "dec_bwd_persist_kernel", "dec_fwd_persist_kernel", "enc_bwd_persist_kernel", "enc_fwd_persist_kernel",
                    "gru_bwd2_diag_kernel", "gru_fwd2_diag_kernel", "gru_bwd2_kernel", "gru_fwd2_kernel", "gemm_tn_kernel",
                    "gemm_nt_kernel", "colsum_kernel", "ce_kernel", "adam_kernel", "sample_head_kernel"):
            if tag in k:
                short = tag
        f_kib = F.get(k, (0, 0.0))[1]
        w_kib = W.get(k, (0, 0.0))[1]
        e = out.setdefault(short, {"launches": 0, "fetch_bytes_raw": 0.0, "write_bytes": 0.0})
        n = max(F.get(k, (0, 0))[0], W.get(k, (0, 0))[0])
        e["fetch_bytes_raw"] += f_kib * 1024 * n
        e["write_bytes"] += w_kib * 1024 * n
        e["launches"] += n
    for e in out.values():
        n = e["launches"]
        e["fetch_bytes_raw"] /= n
        e["write_bytes"] /= n
        e["fetch_bytes_corrected"] = 2.0 * e["fetch_bytes_raw"]
        e["hbm_bytes_per_launch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
    with open(out_json, "w") as f:
        json.dump({"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 2 "
                           "--warmup 1 --no_cpu_baseline --no_roofline`; per-launch averages; FETCH_SIZE doubled "
                           "(gfx950 correction, MI355X_MICROARCH.md); memory-side-of-L2 counters, Infinity-Cache hits included",
                   dtype: out}, f, indent=1, sort_keys=True)


def counters(dbs, out_json, note):
    """Per-kernel averages (per launch, summed over the device's shader engines as rocprofv3 reports them) of every counter
    found in the given --pmc result databases; kernels matched by the names of interest only."""
    tags = ("dec_sample_persist_kernel", "dec_bwd_persist_kernel", "dec_fwd_persist_kernel", "enc_bwd_persist_kernel", "enc_fwd_persist_kernel",
            "gemm_tn256_kernel", "tn_reduce_kernel", "gemm_tn_kernel", "gemm_nt_kernel", "fc_ce_kernel", "adam_kernel")
    out = {}
    for db in dbs:
        c = sqlite3.connect(db)
        for kname, cname, n, v in c.execute(
                "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            for tag in tags:
                if tag in kname:
                    e = out.setdefault(tag, {})
                    e[cname] = v
                    e["launches_" + cname] = n
                    break
    for e in out.values():   # derived ratios (MI355X_MICROARCH.md counter table)
        if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
            w = e["SQ_WAVE_CYCLES"]
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
                if k in e:
                    e["frac_of_wave_cycles_" + k] = e[k] / w
        # SQ_WAVE_CYCLES counts quad-cycles (4 clocks); the persistent kernels hold ONE wave per SIMD, so 4 * WAVE_CYCLES is
        # their SIMD time and MFMA_BUSY / that is the matrix-pipe utilisation.  Not valid at higher occupancy.
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_WAVE_CYCLES"):
            e["mfma_util_if_one_wave_per_simd"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * e["SQ_WAVE_CYCLES"])
    with open(out_json, "w") as f:
        json.dump({"_how": note, "bf16": out}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(sys.argv[4:], sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --kernel-trace --stats")
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
