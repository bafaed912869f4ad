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
        if "PG_EPI_WAIT" in t:
            ins.append("; PG_EPI_WAIT")
            continue
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
            if p.startswith(";"):
                continue
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
    # asm-issued epilogue loads (PG_EPI_LOAD) must not have their destination registers touched before PG_EPI_WAIT
    pending, touched, nload = set(), 0, 0
    for t in ins:
        if t.startswith("s_endpgm"):
            pending = set()
        if "PG_EPI_LOAD" in t:
            pending |= regs(t.split(None, 1)[1].split(",")[0].strip())
            nload += 1
            continue
        if "PG_EPI_WAIT" in t:
            pending = set()
            continue
        if pending and not t.startswith(("s_", ";")):
            body = t.split(";")[0]
            used = set()
            for tok in re.findall(r"[av]\[\d+:\d+\]|\b[av]\d+\b", body):
                used |= regs(tok)
            if used & pending:
                print("epilogue load register touched before its wait: %s" % t)
                touched += 1
    print("%s: %d epilogue loads, %d early touches" % (path, nload, touched))
    return 1 if bad or touched else 0

if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
