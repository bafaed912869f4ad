"""Print the body of one loop of a hipcc -S listing (label -> its backward branch).  Usage: python tools/loop_dump.py file.s .LBB0_76 [kernel-substring]"""
import re
import sys

def main(path, label, kern_sub=""):
    lines = open(path).read().split("\n")
    start = None
    kern = ""
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S+:", l):
            kern = l
        if l.startswith(label + ":") and kern_sub in kern:
            start = i
            break
    for i in range(start + 1, len(lines)):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\S+)", lines[i])
        if m and m.group(1) == label:
            end = i
            break
    for l in lines[start:end + 1]:
        t = l.strip()
        if t and not t.startswith(";"):
            print(l.split(";")[0].rstrip())

if __name__ == "__main__":
    main(*sys.argv[1:])
