#!/usr/bin/env python3
"""Generator of the tile bodies of the one-wavefront-per-SIMD head-dim-16 attention kernels
(ranked-list-truncation_amd/csrc/attention6n_{dq1,dkv1}_body.inc; kernels in csrc/attention6n.hip).

A tile body is NS slots; a slot carries the MFMAs of four pipeline stages of four different ITEMS (an item = one 32-row block of
the tile x one 16-row block of the wavefront's own rows = 512 scores):

    dq   (22 MFMAs per slot):  R1(s-2) x2 | dQ(s-3) x6 | R2(s-2) x2 | S(s) x6 | dP(s) x6
    dkv  (32 MFMAs per slot):  RP1(s-2) x2, RD1(s-2) x2 | dV(s-3) x6, dK(s-3) x6 | RP2(s-2) x2, RD2(s-2) x2 | S(s) x6 | dP(s) x6

(S, dP: the row products; R*: the matrix-pipe residuals of the three-way split of dS (and P); dQ / dV / dK: the list-contracted
outputs, five small plane products into the second accumulator, h h' into the first.)  Every MFMA is followed by a "gap" that the
same wavefront fills with vector work and LDS reads - a v_mfma_f32_16x16x32_bf16 hides about 8 cycles of vector issue behind
it (tools/micro/mfma16_gap.hip: two plain instructions, or one v_exp_f32, or a conversion + one plain).  This script places the
element-wise chunks (exp2, P (dP - delta) or P dP, the v_cvt_pk_bf16_f32 of the three planes), the LDS fragment reads of the next
32-row block and the staging of the next tile (LDS-DMA pieces of the pre-split tile records) into the gaps: list scheduling,
earliest deadline first, on the CYCLIC timeline of the tile body (the pipeline runs across tile boundaries: work of the last
three items of a tile sits in the first slots of the next body, on the same ring of registers), with the dependences between
MFMA results, chunks and MFMA operands - including the write-after-read ones of the register rings - checked at generation
time.  One line of calls per gap; GAP_END (sched_barrier) keeps hipcc from reordering.

    python tools/gen_attn6n_body.py dq  > ranked-list-truncation_amd/csrc/attention6n_dq1_body.inc
    python tools/gen_attn6n_body.py dkv > ranked-list-truncation_amd/csrc/attention6n_dkv1_body.inc
    python tools/gen_attn6n_body.py fwd > ranked-list-truncation_amd/csrc/attention6n_fwd1_body.inc

fwd (16 MFMAs per slot):  R1(s-2) x2 | O(s-3) x6 | R2(s-2) x2 | S(s) x6 - the forward pass with a FIXED reference per query (seeded
scores): exp2 of the scores, their sum into the row's normaliser (e_sum; items of the drain tile multiply by a zero flag), the split
of P on the matrix pipe, O^T += V^T P^T.  Chunks of the last items of a tile that land in the next body are emitted as e_exp_p / e_sum_p:
the previous tile is always a live one - except in the first body, where there is none and e_exp_p must produce a zero weight.
"""
import os
import sys

MODE = sys.argv[1] if len(sys.argv) > 1 else "dq"
OMIT = set(filter(None, os.environ.get("GEN_OMIT", "").split(",")))     # timing experiments: leave out the calls of these names (wrong results)
NB = 4                       # 16-row blocks of own rows per wavefront
NB32 = 4                     # 32-row blocks per tile (tile = 128 rows)
NS = NB * NB32               # slots (= items) per tile body
RING = 4                     # item register sets
LAG = 2                      # gaps between an MFMA and the first vector read of its result
MARGIN = 2                   # a chunk that writes an MFMA operand sits at least this many gaps ahead of the MFMA
WAR = 2                      # a fragment register is rewritten at the earliest this many gaps behind the last MFMA that reads it
RD_AHEAD = 10                # an LDS read is issued at least this many gaps (~100 cycles) ahead of the MFMA that takes its data
CAP = 8                      # vector-issue cycles a gap takes before the scheduler looks for another one
COST = {"exp": 8, "mul": 4, "sub": 4, "cvt": 5, "rd": 2, "ld": 3, "sp": 16, "st": 8, "tb": 3}

# ---- slot layout: list of (stage, index within stage); stage -> item offset
if MODE == "fwd":
    LAYOUT = [("R1", k) for k in range(2)] + [("O", k) for k in range(6)] + [("R2", k) for k in range(2)] + [("S", k) for k in range(6)]
    MATS = ("P",)            # fresh operand that is split: P
elif MODE == "dq":
    LAYOUT = [("R1", k) for k in range(2)] + [("O", k) for k in range(6)] + [("R2", k) for k in range(2)] + \
             [("S", k) for k in range(6)] + [("D", k) for k in range(6)]
    MATS = ("D",)            # fresh operands that are split: dS
else:
    LAYOUT = [("RP1", k) for k in range(2)] + [("RD1", k) for k in range(2)] + [("OV", k) for k in range(6)] + \
             [("OK", k) for k in range(6)] + [("RP2", k) for k in range(2)] + [("RD2", k) for k in range(2)] + \
             [("S", k) for k in range(6)] + [("D", k) for k in range(6)]
    MATS = ("P", "D")
OFFSET = {"S": 0, "D": 0, "R1": 2, "R2": 2, "RP1": 2, "RD1": 2, "RP2": 2, "RD2": 2, "O": 3, "OV": 3, "OK": 3}
GS = len(LAYOUT)             # gaps per slot
G = GS * NS                  # gaps per body
POS = {st_k: g for g, st_k in enumerate(LAYOUT)}


def gap_of(stage, k, item):
    """absolute gap (item 0's S stage starts in slot 0) of MFMA k of `stage` of `item`"""
    return (item + OFFSET[stage]) * GS + POS[(stage, k)]


class Task:
    def __init__(self, name, chunks, release, deadline):
        self.name, self.chunks, self.release, self.deadline = name, chunks, release, deadline    # chunks: (call, kind)
        self.after = []      # (task, lag): first chunk at gap >= task.done + lag
        self.placed = []


tasks = []
used = [0] * G
sched = [[] for _ in range(G)]


def add(name, chunks, release, deadline, after=()):
    t = Task(name, chunks, release, deadline)
    t.after = list(after)
    tasks.append(t)
    return t


for i in range(NS):
    it, n, b32 = i % RING, i % NB, i // NB
    s_done = [gap_of("S", 2, i), gap_of("S", 5, i)]
    d_done = [gap_of("D", 2, i), gap_of("D", 5, i)] if MODE != "fwd" else None
    # element-wise: per register (kb, r)
    ep, ed, es = {}, {}, {}
    for kb in range(2):
        for r in range(4):
            ep[kb, r] = add(f"ep{i}.{kb}{r}", [(f"e_exp({it}, {kb}, {r});", "exp")], s_done[kb] + LAG, None)
            if MODE == "fwd":      # the weight into the normaliser of own block n (any time before the registers turn into residuals)
                es[kb, r] = add(f"es{i}.{kb}{r}", [(f"e_sum({it}, {n}, {kb}, {r});", "mul")], 0, gap_of("R1", kb, i) - 1, [(ep[kb, r], 1)])
            else:
                ed[kb, r] = add(f"ed{i}.{kb}{r}", [(f"e_mul({it}, {kb}, {r});", "mul")], d_done[kb] + LAG, None, [(ep[kb, r], 1)])
    for m in MATS:
        src = ep if m == "P" else ed
        r1 = "R1" if MODE in ("dq", "fwd") else f"R{m}1"
        r2 = "R2" if MODE in ("dq", "fwd") else f"R{m}2"
        out = ("O" if MODE in ("dq", "fwd") else ("OV" if m == "P" else "OK"))
        mi = 0 if m == "P" else 1
        c0 = []
        for j in range(4):
            kb, rr = j >> 1, j & 1
            t = add(f"c0{m}{i}.{j}", [(f"c_pk({it}, {mi}, 0, {j});", "cvt")], 0, gap_of(r1, 0, i) - MARGIN,
                    [(src[kb, 2 * rr], 1), (src[kb, 2 * rr + 1], 1)])
            c0.append(t)
            # (the dS registers are rewritten by the exp of ... no: P and dS live in their own registers)
        for j in range(4):
            kb = j >> 1
            add(f"c1{m}{i}.{j}", [(f"c_pk({it}, {mi}, 1, {j});", "cvt")], gap_of(r1, kb, i) + LAG, gap_of(r2, 0, i) - MARGIN)
        for j in range(4):
            kb = j >> 1
            # plane l is first read by the third product of the output stage; the item's fp32 registers are rewritten by the score
            # product of item i + RING
            dl = min(gap_of(out, 2, i) - MARGIN, gap_of("S", 0, i + RING) - 1)
            add(f"c2{m}{i}.{j}", [(f"c_pk({it}, {mi}, 2, {j});", "cvt")], gap_of(r2, kb, i) + LAG, dl)
        # deadlines of the element-wise chunks follow from the conversions that read them
    for kb in range(2):
        for r in range(4):
            j = 2 * kb + (r >> 1)
            if MODE == "fwd":
                ep[kb, r].deadline = gap_of("R1", 0, i) - MARGIN - 2
                continue
            if "P" in MATS:
                ep[kb, r].deadline = gap_of("RP1", 0, i) - MARGIN - 1
            ed[kb, r].deadline = gap_of("R1" if MODE == "dq" else "RD1", 0, i) - MARGIN - 1
            if "P" not in MATS:
                ep[kb, r].deadline = ed[kb, r].deadline - 1

# ---- LDS fragment reads of 32-row block b32 into fragment buffer b32 & 1: any time after the last MFMA that uses block b32 - 2
# ---- (the previous tenant of the buffer; not before the barrier that opens the tile), RD_AHEAD gaps ahead of the first MFMA
# ---- that uses this one - a window of a whole 32-row block, so the reads spread out instead of arriving in one burst (the
# ---- single-buffered form made the wavefront wait for every burst: 26 % of the dQ kernel, profiles/r05_notes.md)
first_item = lambda b32: b32 * NB
last_item = lambda b32: b32 * NB + NB - 1
for b32 in range(NB32):
    fb = b32 & 1
    prev_last = last_item(b32 - 2) if b32 >= 2 else None
    for mat, stage in ((("k", "S"),) if MODE == "fwd" else (("k" if MODE == "dq" else "q", "S"), ("v" if MODE == "dq" else "d", "D"))):
        for kb in range(2):
            for w in range(3):
                use = gap_of(stage, 3 * kb + w, first_item(b32))
                rel = gap_of(stage, 3 * kb + w, prev_last) + WAR if prev_last is not None else 0
                add(f"r{mat}{b32}.{kb}{w}", [(f"rd_row({fb}, {0 if stage == 'S' else 1}, {kb}, {w}, {b32});", "rd")], rel, use - RD_AHEAD)
    outs = (("O", 1),) if MODE == "fwd" else (("O", 0),) if MODE == "dq" else (("OV", 1), ("OK", 0))      # fwd: V^T; dq: K^T; dkv: dO^T for dV, Q^T for dK
    for stage, mat in outs:
        # planes in the order the six products take them (A operand): m, l, h, m, h, h -> first uses k = 0 (m), 1 (l), 2 (h);
        # last uses k = 3 (m), 1 (l), 5 (h)
        for pl, (fu, lu) in (("m", (0, 3)), ("l", (1, 1)), ("h", (2, 5))):
            pi = {"h": 0, "m": 1, "l": 2}[pl]
            for half in range(2):
                use = gap_of(stage, fu, first_item(b32))
                # previous tenant of the buffer: block b32 - 2 (of the PREVIOUS TILE for b32 < 2: its output stage sits up to three
                # slots into this body - the timeline is cyclic)
                rel = max(0, gap_of(stage, lu, last_item(b32 - 2) if b32 >= 2 else last_item(b32 + NB32 - 2) - NS) + WAR)
                add(f"t{mat}{b32}.{pl}{half}", [(f"rd_tr({fb}, {mat}, {pi}, {half}, {b32});", "rd")], rel, use - RD_AHEAD)
    if MODE == "dkv":
        # lse / delta seeds of the block's 32 rows (the C operands of S / dP): [kb] float4 each
        for kb in range(2):
            for which, stage in ((0, "S"), (1, "D")):
                use = gap_of(stage, 3 * kb, first_item(b32))
                rel = gap_of(stage, 3 * kb, prev_last) + WAR if prev_last is not None else 0
                add(f"tb{b32}.{kb}{which}", [(f"rd_tab({fb}, {which}, {kb}, {b32});", "tb")], rel, use - RD_AHEAD)

# ---- staging of the next tile: LDS-DMA pieces of the pre-split tile records (12 + 12 (+ 1: the seeds of dK+dV) pieces of 1 KiB, a
# ---- wavefront issues every fourth one): early in the body, so that they have landed long before the barrier
for j in range(7 if MODE == "dkv" else 6):
    add(f"dma{j}", [(f"st_dma({j});", "ld")], GS + 3 * j, GS * 6)

def place(t, capv):
    lo = t.release
    for dep, lag in t.after:
        assert dep.placed, (t.name, dep.name)
        lo = max(lo, dep.placed[-1] + lag)
    g = lo
    for call, kind in t.chunks:
        c = COST[kind]
        while used[g % G] + c > max(capv[g % G], c) or (kind == "exp" and any(x[1] == "exp" for x in sched[g % G])):
            g += 1
            if t.deadline is not None and g > t.deadline:
                return False
        used[g % G] += c
        # (forward: chunks of a tile's last items that land in the NEXT body get their own names - in the first body they belong to no tile)
        sched[g % G].append((call.replace("e_sum(", "e_sum_p(").replace("e_exp(", "e_exp_p(") if (MODE == "fwd" and g >= G) else call, kind, t.name))
        t.placed.append(g)
        g += 1
    return t.deadline is None or t.placed[-1] <= t.deadline


def run(capv):
    for g in range(G):
        used[g] = 0
        sched[g] = []
    for t in tasks:
        t.placed = []
    # dependency-respecting EDF: repeatedly take the unplaced task with the earliest deadline whose dependences are placed
    pending = list(tasks)
    while pending:
        ready = [t for t in pending if all(d.placed for d, _ in t.after)]
        t = min(ready, key=lambda t: (t.deadline if t.deadline is not None else 1 << 30, t.release))
        if not place(t, capv):
            return t
        pending.remove(t)
    return None


capv = [CAP] * G
for attempt in range(400):
    miss = run(capv)
    if miss is None:
        break
    lo = miss.release
    for dep, lag in miss.after:
        lo = max(lo, (dep.placed[-1] if dep.placed else 0) + lag)
    hi = miss.deadline if miss.deadline is not None else lo + GS
    for g in range(min(lo, hi) - 6, hi + 1):
        capv[g % G] += 1
else:
    sys.exit(f"no schedule: {miss.name} (release {miss.release}, deadline {miss.deadline})")
cap = max(capv)
sys.stderr.write(f"{MODE}: {G} gaps, capacity {CAP}..{cap} cycles per gap, mean load {sum(used) / G:.1f}, max {max(used)}, "
                 f"{sum(1 for u in used if u > 8)} gaps over 8 cycles (sum of the excess {sum(max(0, u - 8) for u in used)})\n")

# ---- emit
CALL = {"S": "m_s", "D": "m_d", "R1": "m_r", "R2": "m_r", "RP1": "m_r", "RD1": "m_r", "RP2": "m_r", "RD2": "m_r", "O": "m_o", "OV": "m_o", "OK": "m_o"}
out = [f"// generated by tools/gen_attn6n_body.py {MODE} - do not edit"]
for s in range(NS):
    out.append(f"// slot {s}")
    out.append(f"A6N_STAMP({s});")
    for g0, (stage, k) in enumerate(LAYOUT):
        i = s - OFFSET[stage]                 # item (negative: of the previous tile - same ring slot, same own-row block)
        it, n = i % RING, i % NB
        fb = (i // NB) & 1                     # fragment buffer of the item's 32-row block (i < 0: blocks 3, 2 of the previous tile)
        if stage in ("S", "D"):
            call = f"{CALL[stage]}({it}, {n}, {k}, {fb});"
        elif stage.startswith("R"):
            which = 0 if MODE == "fwd" else (1 if stage in ("R1", "R2", "RD1", "RD2") else 0)
            call = f"m_r({it}, {which}, {1 if stage.endswith('1') else 2}, {k});"
        else:
            which = 0 if MODE == "fwd" else {"O": 1, "OV": 0, "OK": 1}[stage]
            call = f"m_o({it}, {n}, {which}, {k}, {fb});"
        work = " ".join(c for c, _k, _n in sched[s * GS + g0] if c.split("(")[0] not in OMIT)
        out.append(f"{call} GAP_END; {work} GAP_END;".replace("  ", " "))
out.append(f"// {sum(1 for x in sched if x)} of {G} gaps carry vector work; capacity {cap} cycles per gap")
print("\n".join(out))
