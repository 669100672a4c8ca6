"""Instruction-class pattern of a kernel's loops in hipcc's device assembly (-S --cuda-device-only).

usage: python tools/isa_pattern.py file.s kernel-substring
Prints, per basic block with at least 8 matrix instructions, the run-length encoded sequence of
M (v_mfma), V (other VALU), D (ds_*), G (global/buffer), S (scalar), W (s_waitcnt), B (s_barrier).
"""
import re
import sys


def klass(op):
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith("v_"):
        return "V"
    if op.startswith("ds_"):
        return "D"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "G"
    if op.startswith("s_waitcnt"):
        return "W"
    if op.startswith("s_barrier"):
        return "B"
    if op.startswith("s_nop"):
        return "n"
    return "S"


def main():
    path, want = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    inside, block, name = False, [], None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            inside = want in m.group(1)
            name = m.group(1)
            if inside:
                print("==", name)
            block, label = [], "entry"
            continue
        if not inside:
            continue
        if ln.startswith(".Lfunc_end"):
            inside = False
            continue
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            emit(label, block)
            block, label = [], m.group(1)
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".")):
            continue
        block.append(klass(t.split()[0]))
    emit(label, block)


def emit(label, block):
    if block.count("M") < 8:
        return
    out, i = [], 0
    while i < len(block):
        j = i
        while j < len(block) and block[j] == block[i]:
            j += 1
        out.append(f"{block[i]}{j - i if j - i > 1 else ''}")
        i = j
    c = {k: block.count(k) for k in "MVDGSWB"}
    print(label, c)
    print("   " + " ".join(out))


if __name__ == "__main__":
    main()


def gap_model(block, mfma=32, hold=8, cost=None):
    """Guide model (MI355X_MICROARCH.md constants): an MFMA gap runs max(32, 8 + sum of the issue costs in the gap)."""
    cost = cost or {"V": 4.3, "D": 4, "S": 1, "W": 0, "B": 0, "n": 4, "G": 8}
    total, cur, seen = 0.0, None, False
    pre = 0.0
    for k in block:
        if k == "M":
            if seen:
                total += max(mfma, hold + cur)
            cur, seen = 0.0, True
        elif seen:
            cur += cost[k]
        else:
            pre += cost[k]
    if seen:
        total += max(mfma, hold + cur)
    return pre + total
