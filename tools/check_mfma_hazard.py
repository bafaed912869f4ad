"""Static check of the hand-written MFMA asm in the persistent kernels: an inline-asm MFMA without its own `s_nop` must not
follow a VALU write (v_accvgpr_write / v_mov / ...) of one of its operands within two instructions - hipcc cannot see the
hazard inside an asm statement.  Usage: python tools/check_mfma_hazard.py file.s  (hipcc --cuda-device-only -S output)"""
import re
import sys

def regs(tok):
    m = re.match(r"([av])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([av])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()

def main(path):
    ins = []
    for line in open(path):
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        ins.append(t)
    bad = 0
    for i, t in enumerate(ins):
        if not t.startswith("v_mfma"):
            continue
        ops = [x.strip() for x in t.split(None, 1)[1].split(",")]
        need = set().union(*[regs(o) for o in ops[1:4]])
        # walk back over at most two issued instructions (an s_nop N counts N + 1 wait states)
        states = 0
        for k in range(i - 1, max(i - 6, -1), -1):
            p = ins[k]
            if p.startswith("s_nop"):
                states += int(p.split()[1]) + 1
                continue
            if states >= 2:
                break
            if p.startswith("v_") and not p.startswith("v_mfma") and not p.startswith("v_cmp"):
                dst = regs(p.split(None, 1)[1].split(",")[0].strip())
                if dst & need:
                    print("hazard: %s  ->  %s" % (p, t))
                    bad += 1
            states += 1
    print("%s: %d MFMA hazards" % (path, bad))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
