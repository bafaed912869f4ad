"""Instruction counts of the innermost loops of every kernel in a hipcc -S listing (one wave per SIMD issues one instruction
at a time, so the tile loops of the persistent kernels are priced in instructions).  Usage: python tools/loop_instrs.py file.s"""
import re
import sys
from collections import Counter

def main(path, top=10, min_mfma=90):
    seen = set()
    lines = open(path).read().split("\n")
    kern = None
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\S+):\s*(;.*)?$", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S+:", l):
            kern = l.split(":")[0]
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\S+)", l) or re.match(r"\s+s_branch\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            body = [x.split()[0] for x in lines[a:i + 1] if x.strip() and not x.strip().startswith((";", ".")) and not x.rstrip().endswith(":")]
            if sum(1 for x in body if x.startswith("v_mfma")) < min_mfma or (kern, m.group(1)) in seen:
                continue
            seen.add((kern, m.group(1)))
            c = Counter(body)
            print("%s\n  loop %s: %d instructions; %s" % (kern[:90], m.group(1), len(body), ", ".join("%s %d" % kv for kv in c.most_common(top))))

if __name__ == "__main__":
    main(sys.argv[1])
