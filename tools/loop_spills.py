"""Per-loop summary of a hipcc -S listing: instructions, MFMAs, scratch (spill) traffic, LDS reads, barriers, AGPR copies.
Usage: python tools/loop_spills.py file.s [min_mfma]"""
import re
import sys
from collections import Counter

def main(path, min_mfma=40):
    lines = open(path).read().split("\n")
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\S+):\s*(;.*)?$", l)
        if m:
            labels[m.group(1)] = i
    seen = set()
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\S+)", l) or re.match(r"\s+s_branch\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            body = [x.split()[0] for x in lines[a:i + 1] if x.strip() and not x.strip().startswith((";", ".")) and not x.rstrip().endswith(":")]
            c = Counter(body)
            nm = sum(v for k, v in c.items() if k.startswith("v_mfma"))
            if nm < int(min_mfma) or m.group(1) in seen:
                continue
            seen.add(m.group(1))
            sl = sum(v for k, v in c.items() if k.startswith("scratch_load"))
            ss = sum(v for k, v in c.items() if k.startswith("scratch_store"))
            print("loop %-10s %5d instrs, mfma %3d, scratch ld/st %3d/%3d, ds_read %3d, ds_write %2d, barriers %d, accvgpr %3d, vmem %d" % (
                m.group(1), len(body), nm, sl, ss, sum(v for k, v in c.items() if k.startswith("ds_read")),
                sum(v for k, v in c.items() if k.startswith("ds_write")), c.get("s_barrier", 0),
                sum(v for k, v in c.items() if k.startswith("v_accvgpr")),
                sum(v for k, v in c.items() if k.startswith(("global_", "buffer_")))))

if __name__ == "__main__":
    main(*sys.argv[1:])
