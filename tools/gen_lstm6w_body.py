#!/usr/bin/env python3
"""Tick bodies of the whole-weights BiLSTM recurrences (csrc/lstm6w.hip): one wavefront per SIMD runs the MFMA chain of one
16-list half while the vector ALU works through the element-wise part of the OTHER half, so every MFMA is followed by a fenced
gap with ~8 cycles of vector / LDS / memory work (what a v_mfma_f32_16x16x32_bf16 hides, profiles/r05_notes.md).

    python tools/gen_lstm6w_body.py fwd|fwd_xin|fwd_seq|fwd_xin_seq|bwd > ranked-list-truncation_amd/csrc/lstm6w_<kind>_body.inc

A body is a flat list of macro calls; the kernel defines them (chain half X, element-wise half Y).  The chain: 4 k-steps
(forward: of the 128 hidden units; backward: 16 k-steps of the 512 gate rows) x the six plane products, smallest first per
accumulator, consecutive MFMAs on different accumulators.  The element-wise work is emitted in dependence order, breadth
first over the four units a lane holds per 16-unit block, so neighbours in the list are independent; the filler gives every gap
up to BUDGET cycles of it.  LDS fragment reads of the chain sit at fixed positions ahead of their first use."""
import sys

BUDGET = int(__import__('os').environ.get('GEN_BUDGET', '8'))
PROD = [("m", "m"), ("l", "h"), ("h", "l"), ("m", "h"), ("h", "m"), ("h", "h")]      # (W plane, operand plane), smallest first


def fwd_elem(xin):
    """Element-wise micro-ops of one half: (text, cycles).  Both 16-unit blocks of the lane advance in lock step, so that the loads /
    stores of the two 64-byte halves of a 128-byte line (block 0 | block 1 of a wavefront's 32 units) are issued back to back."""
    ops = []
    U = (0, 1)
    if not xin:
        for g in range(4):
            for u in U:
                ops.append((f"EA({u}, {g})", 8))           # acc += pre-activation
        for g in range(4):
            for u in U:
                ops.append((f"LG({u}, {g})", 4))           # fetch the next step's pre-activations into the freed registers
    for g in range(4):
        for u in U:
            ops.append((f"ES({u}, {g})", 8))               # scale for exp2
    for g in range(4):
        for u in U:
            for r in range(4):
                ops.append((f"EX({u}, {g}, {r})", 8))
    for g in range(4):
        for u in U:
            ops.append((f"E1({u}, {g})", 8))
    for g in range(4):
        for u in U:
            for r in range(4):
                ops.append((f"ER({u}, {g}, {r})", 8))
    for u in U:
        ops.append((f"EG({u})", 8))                        # tanh of the cell candidate from its sigmoid form
    for g in range(4):
        for u in U:
            ops.append((f"SG({u}, {g})", 4))               # store the activated gates
    for name, cyc in (("EC1", 8), ("EC2", 8), ("SC", 4), ("ET", 8)):
        for u in U:
            ops.append((f"{name}({u})", cyc))
    for u in U:
        for r in range(4):
            ops.append((f"EXC({u}, {r})", 8))
    for u in U:
        ops.append((f"E1C({u})", 8))
    for u in U:
        for r in range(4):
            ops.append((f"ERC({u}, {r})", 8))
    for name, cyc in (("EH1", 8), ("EH2", 8), ("SH", 4)):
        for u in U:
            ops.append((f"{name}({u})", cyc))
    for u in U:
        for part, cyc in enumerate((8, 16, 16, 8, 16, 16, 8)):
            ops.append((f"SP({u}, {part})", cyc))          # three-way split of the four new h values, in seven pieces
        for pl in range(3):
            ops.append((f"LW({u}, {pl})", 4))
    if xin:
        for part in range(4):
            ops.append((f"XB({part})", 8))                 # the input-projection operand of this half's next chain + the next x fetch
    return ops


def fwd_elem_seq(xin):
    """... block after block (the single-half kernels: a step is latency there, and the shorter dependence distance of this order wins)."""
    ops = []
    for u in (0, 1):
        if not xin:
            for g in range(4):
                ops.append((f"EA({u}, {g})", 8))
            for g in range(4):
                ops.append((f"LG({u}, {g})", 4))
        for g in range(4):
            ops.append((f"ES({u}, {g})", 8))
        for g in range(4):
            for r in range(4):
                ops.append((f"EX({u}, {g}, {r})", 8))
        for g in range(4):
            ops.append((f"E1({u}, {g})", 8))
        for g in range(4):
            for r in range(4):
                ops.append((f"ER({u}, {g}, {r})", 8))
        ops.append((f"EG({u})", 8))
        for g in range(4):
            ops.append((f"SG({u}, {g})", 4))
        ops.append((f"EC1({u})", 8))
        ops.append((f"EC2({u})", 8))
        ops.append((f"SC({u})", 4))
        ops.append((f"ET({u})", 8))
        for r in range(4):
            ops.append((f"EXC({u}, {r})", 8))
        ops.append((f"E1C({u})", 8))
        for r in range(4):
            ops.append((f"ERC({u}, {r})", 8))
        ops.append((f"EH1({u})", 8))
        ops.append((f"EH2({u})", 8))
        ops.append((f"SH({u})", 4))
        for part, cyc in enumerate((8, 16, 16, 8, 16, 16, 8)):
            ops.append((f"SP({u}, {part})", cyc))
        for pl in range(3):
            ops.append((f"LW({u}, {pl})", 4))
    if xin:
        for part in range(4):
            ops.append((f"XB({part})", 8))
    return ops


def fwd_body(xin, seq=False):
    mf = []
    if xin:
        for rb in range(8):
            mf.append(f"MX({rb})")
    for ks in range(4):
        for rq in range(2):
            for p, (wp, bp) in enumerate(PROD):
                for j in range(4):
                    rb = 4 * rq + j
                    first = (not xin) and ks == 0 and p == 0
                    mf.append(f"MF({rb}, {ks}, {p}, {1 if first else 0})")
    nx = 8 if xin else 0
    fixed = {}                                             # gap index -> LDS reads of the chain
    for G in range(8):
        ks, rq = divmod(G, 2)
        base = nx + 24 * G
        for j in range(4):                                 # l fragments of the next group (the last group: of the next tick's first)
            Gn = (G + 1) % 8
            fixed.setdefault(base + 8 + j, []).append(f"RL({4 * (Gn % 2) + j}, {Gn // 2}, {j})")
        if rq == 1 and ks < 3:
            for pl in range(3):
                fixed.setdefault(base + 12 + pl, []).append(f"RB({ks + 1}, {pl})")
    ops = fwd_elem_seq(xin) if seq else fwd_elem(xin)
    out = []
    qi = 0
    for i, m in enumerate(mf):
        line = [m]
        budget = BUDGET
        for f in fixed.get(i, []):
            line.append(f)
            budget -= 4
        while qi < len(ops) and (ops[qi][1] <= budget or budget == BUDGET):
            line.append(ops[qi][0])
            budget -= ops[qi][1]
            qi += 1
        out.append("; ".join(line) + "; GAP_END;")
    rest = []
    while qi < len(ops):
        rest.append(ops[qi][0])
        qi += 1
    if rest:
        out.append("; ".join(rest) + "; GAP_END;")
    return out


def bwd_elem(paired=True):
    """Element-wise micro-ops of one half of the backward step (csrc/lstm6w.hip names them).  paired: the dA stores and the gate reloads
    of the two unit blocks (the 64-byte halves of a 128-byte line) back to back - both blocks' dA first, then the stores, the splits, the
    reloads; otherwise block after block."""
    arith = (("B1", 8), ("LDN10", 4), ("B2", 8), ("B3", 16), ("B5", 16), ("B6", 8), ("B7", 16), ("B8", 8), ("ST3", 4),
             ("B9", 8), ("B10", 16), ("B11", 16), ("ST0", 4), ("B12", 16), ("B13", 8), ("ST2", 4), ("B14", 8), ("B15", 16),
             ("LDN8", 4), ("B16", 16), ("B17", 8), ("B18", 8), ("ST1", 4))

    def tanh_ops(u):
        o = [(f"T1({u})", 8)]
        o += [(f"TX({u}, {r})", 8) for r in range(4)]
        o.append((f"T3({u})", 8))
        o += [(f"TR({u}, {r})", 8) for r in range(4)]
        o.append((f"T5({u})", 16))
        return o

    def split_ops(u):
        o = []
        for ch in (0, 1):
            o += [(f"SP({u}, {ch}, {part})", cyc) for part, cyc in enumerate((16, 24, 24, 16, 24, 24, 16))]
            o += [(f"LW({u}, {ch}, {pl})", 4) for pl in range(3)]
        return o

    ops = []
    if not paired:
        for u in (0, 1):
            ops += tanh_ops(u)
            for name, cyc in arith:
                if name.startswith("LDN"):
                    ops.append((f"LDN({int(name[3:]) + u})", cyc))
                elif name.startswith("ST"):
                    ops.append((f"ST({u}, {name[2:]})", cyc))
                else:
                    ops.append((f"{name}({u})", cyc))
            ops += split_ops(u)
            ops += [(f"LDN({4 * u + g})", 4) for g in range(4)]
        return ops
    for u in (0, 1):
        ops += tanh_ops(u)
        for name, cyc in arith:
            if name.startswith("LDN"):
                ops.append((f"LDN({int(name[3:]) + u})", cyc))
            elif not name.startswith("ST"):
                ops.append((f"{name}({u})", cyc))
    for g in (3, 0, 2, 1):
        for u in (0, 1):
            ops.append((f"ST({u}, {g})", 4))
    for u in (0, 1):
        ops += split_ops(u)
    for g in range(4):
        for u in (0, 1):
            ops.append((f"LDN({4 * u + g})", 4))
    return ops


def bwd_body(paired=True):
    mf = []
    for ks in range(16):
        for p in range(6):
            for ub in range(2):
                mf.append(f"MB({ub}, {ks}, {p}, {1 if ks == 0 and p == 0 else 0})")
    fixed = {}
    for ks in range(16):
        base = 12 * ks
        if ks < 15:
            for pl in range(3):
                fixed.setdefault(base + 5 + pl, []).append(f"RB({ks + 1}, {pl})")
        if 7 <= ks < 15:                                   # the l fragments of k-steps 8-15 come from LDS, one k-step ahead
            for ub in range(2):
                fixed.setdefault(base + 9 + ub, []).append(f"RL({ub}, {ks + 1})")
    ops = bwd_elem(paired)
    return fill(mf, fixed, ops)


def fill(mf, fixed, ops):
    out = []
    qi = 0
    for i, m in enumerate(mf):
        line = [m]
        budget = BUDGET
        for f in fixed.get(i, []):
            line.append(f)
            budget -= 4
        while qi < len(ops) and (ops[qi][1] <= budget or budget == BUDGET):
            line.append(ops[qi][0])
            budget -= ops[qi][1]
            qi += 1
        out.append("; ".join(line) + "; GAP_END;")
    rest = [o[0] for o in ops[qi:]]
    if rest:
        out.append("; ".join(rest) + "; GAP_END;")
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    if kind in ("fwd", "fwd_xin", "fwd_seq", "fwd_xin_seq"):
        lines = fwd_body(kind.startswith("fwd_xin"), kind.endswith("_seq"))
    elif kind in ("bwd", "bwd_seq"):
        lines = bwd_body(kind == "bwd")
    else:
        raise SystemExit("kind: fwd | fwd_xin | fwd_seq | fwd_xin_seq | bwd")
    print(f"// generated by tools/gen_lstm6w_body.py {kind} - do not edit")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
