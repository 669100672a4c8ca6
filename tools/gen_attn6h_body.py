#!/usr/bin/env python3
"""Generator of the tile bodies of the one-wavefront-per-SIMD head-dim-64 attention kernels on v_mfma_f32_16x16x32_bf16
(ranked-list-truncation_amd/csrc/attention6h_{fwd1,dq1,dkv1}_body.inc; kernels in csrc/attention6h.hip).

The form is that of the head-dim-16 kernels (tools/gen_attn6n_body.py): a tile body is NS slots, a slot carries the MFMAs of the
pipeline stages of different ITEMS (an item = one 32-row block of the 64-row tile x one 16-row block of the wavefront's own rows =
512 scores), every MFMA is followed by a "gap" that the same wavefront fills with vector work (exp2, row sums, P dP, the
v_cvt_pk_bf16_f32 of the three planes), the LDS fragment reads of the next 32-row block and the LDS-DMA staging of the next tile.
What head dim 64 changes: a d-contracted product is two k-steps of each of the six plane products (12 MFMAs per 16 x 16 tile, 24 per
item), a d-indexed output four 16-row blocks (24 MFMAs per item), so a slot is 52 (forward) / 76 (dQ) / 104 (dK+dV) MFMAs long and
the vector work of an item (the same as at head dim 16) fills a third of the gaps' capacity; the fragment registers of a 32-row block
are SINGLE-buffered (a register is reloaded in the gaps between its last use by own block NB - 1 and its first use by own block 0 of
the next 32-row block: at least a quarter of a slot).

    fwd (NB = 4, 52 per slot):  R1(s-2) x2 | O(s-3) x24 | R2(s-2) x2 | S(s) x24
    dq  (NB = 2, 76 per slot):  R1(s-2) x2 | O(s-3) x24 | R2(s-2) x2 | S(s) x24 | D(s) x24
    dkv (NB = 2, 104 per slot): RP1, RD1 (s-2) x2 each | OV(s-3) x24 | OK(s-3) x24 | RP2, RD2 (s-2) x2 each | S(s) x24 | D(s) x24

(S, D: the row products S = Q K^T and dP = dO V^T, index k = 12 kb + 2 product + k-step; R*: the matrix-pipe residuals of the
three-way split; O*: the list-contracted outputs, index k = 4 product + d block, the five small plane products into the second
accumulator.)  List scheduling, earliest deadline first, on the cyclic timeline of the tile body; dependences between MFMA results,
chunks and MFMA operands - including the write-after-read ones of the register rings and of the single-buffered fragments - are
checked at generation time.  One line of calls per gap; GAP_END (sched_barrier) keeps hipcc from reordering.

    python tools/gen_attn6h_body.py fwd > ranked-list-truncation_amd/csrc/attention6h_fwd1_body.inc
    python tools/gen_attn6h_body.py dq  > ranked-list-truncation_amd/csrc/attention6h_dq1_body.inc
    python tools/gen_attn6h_body.py dkv > ranked-list-truncation_amd/csrc/attention6h_dkv1_body.inc

fwd: the weights are exp2(score - fixed reference of the query) (seeded scores), summed into the row's normaliser (e_sum; items of
the drain tile multiply by a zero flag).  Chunks of the last items of a tile that land in the next body are emitted as e_exp_p /
e_sum_p: the previous tile is always a live one - except in the first body, where there is none and e_exp_p must produce zero.
"""
import os
import sys

MODE = sys.argv[1] if len(sys.argv) > 1 else "fwd"
DROP = len(sys.argv) > 2 and sys.argv[2] == "drop"
OMIT = set(filter(None, os.environ.get("GEN_OMIT", "").split(",")))     # timing experiments: leave out the calls of these names (wrong results)
NB = 4 if MODE == "fwd" else 2      # 16-row blocks of own rows per wavefront
NB32 = 2                     # 32-row blocks per tile (tile = 64 rows)
NS = NB * NB32               # slots (= items) per tile body
RING = 4                     # item register sets
LAG = 2                      # gaps between an MFMA and the first vector read of its result
MARGIN = 2                   # a chunk that writes an MFMA operand sits at least this many gaps ahead of the MFMA
WAR = 2                      # a fragment register is rewritten at the earliest this many gaps behind the last MFMA that reads it
RD_AHEAD = 10                # an LDS read is issued at least this many gaps (~160 cycles) ahead of the MFMA that takes its data
CAP = 8                      # vector-issue cycles a gap takes before the scheduler looks for another one
COST = {"exp": 8, "mul": 4, "sub": 4, "cvt": 5, "rd": 2, "ld": 3, "tb": 3, "drop": 16, "hc": 48}

# the six plane products, smallest first: (plane of the A operand, plane of the B operand); 0 = h, 1 = m, 2 = l
PROD = [(1, 1), (2, 0), (0, 2), (1, 0), (0, 1), (0, 0)]

# ---- slot layout: list of (stage, index within stage); stage -> item offset
def stage(name, n):
    return [(name, k) for k in range(n)]

if MODE == "fwd":
    LAYOUT = stage("R1", 2) + stage("O", 24) + stage("R2", 2) + stage("S", 24)
    MATS = ("P",)            # fresh operand that is split: P
elif MODE == "dq":
    LAYOUT = stage("R1", 2) + stage("O", 24) + stage("R2", 2) + stage("S", 24) + stage("D", 24)
    MATS = ("D",)            # dS
else:
    LAYOUT = stage("RP1", 2) + stage("RD1", 2) + stage("OV", 24) + stage("OK", 24) + stage("RP2", 2) + stage("RD2", 2) + \
             stage("S", 24) + stage("D", 24)
    MATS = ("P", "D")
OFFSET = {"S": 0, "D": 0, "R1": 2, "R2": 2, "RP1": 2, "RD1": 2, "RP2": 2, "RD2": 2, "O": 3, "OV": 3, "OK": 3}
GS = len(LAYOUT)             # gaps per slot
G = GS * NS                  # gaps per body
POS = {st_k: g for g, st_k in enumerate(LAYOUT)}


def gap_of(stg, k, item):
    """absolute gap (item 0's S stage starts in slot 0) of MFMA k of stage `stg` of `item`"""
    return (item + OFFSET[stg]) * GS + POS[(stg, k)]


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


def r_stage(m, lvl):
    return f"R{lvl}" if MODE in ("dq", "fwd") else f"R{m}{lvl}"


def o_stage(m):
    return "O" if MODE in ("dq", "fwd") else ("OV" if m == "P" else "OK")


for i in range(NS):
    it, n, b32 = i % RING, i % NB, i // NB
    s_done = [gap_of("S", 11, i), gap_of("S", 23, i)]
    d_done = [gap_of("D", 11, i), gap_of("D", 23, i)] if MODE != "fwd" else None
    ep, ed, es, ek = {}, {}, {}, {}
    for kb in range(2):
        for r in range(4):
            ep[kb, r] = add(f"ep{i}.{kb}{r}", [(f"e_exp({it}, {kb}, {r});", "exp")], s_done[kb] + LAG, None)
            if MODE == "fwd":      # the weight into the normaliser of own block n (any time before the registers turn into residuals)
                es[kb, r] = add(f"es{i}.{kb}{r}", [(f"e_sum({it}, {n}, {kb}, {r});", "mul")], 0, gap_of("R1", kb, i) - 1, [(ep[kb, r], 1)])
                if DROP:           # train mode: the dropout mask acts on the weight AFTER it went into the normaliser
                    ek[kb, r] = add(f"ek{i}.{kb}{r}", [(f"e_drop({it}, {n}, {kb}, {r}, {b32 & 1});", "drop")], 0, gap_of("R1", 0, i) - MARGIN - 1,
                                    [(es[kb, r], 1)])
            else:
                ed[kb, r] = add(f"ed{i}.{kb}{r}", [(f"e_mul({it}, {kb}, {r});", "mul")], d_done[kb] + LAG, None, [(ep[kb, r], 1)])
    for m in MATS:
        src = (ek if DROP and MODE == "fwd" else ep) if m == "P" else ed
        r1, r2, out = r_stage(m, 1), r_stage(m, 2), o_stage(m)
        mi = 0 if m == "P" else 1
        for j in range(4):
            kb, rr = j >> 1, j & 1
            add(f"c0{m}{i}.{j}", [(f"c_pk({it}, {mi}, 0, {j});", "cvt")], 0, gap_of(r1, 0, i) - MARGIN,
                [(src[kb, 2 * rr], 1), (src[kb, 2 * rr + 1], 1)])
        for j in range(4):
            kb = j >> 1
            add(f"c1{m}{i}.{j}", [(f"c_pk({it}, {mi}, 1, {j});", "cvt")], gap_of(r1, kb, i) + LAG, gap_of(r2, 0, i) - MARGIN)
        # plane l is first read by the output products with B plane 2 (product index 2: k = 8 .. 11); the item's fp32 registers are
        # rewritten by the first row product of item i + RING
        first_l = min(4 * p for p, (_a, b) in enumerate(PROD) if b == 2)
        for j in range(4):
            kb = j >> 1
            dl = min(gap_of(out, first_l, i) - MARGIN, gap_of("S", 0, i + RING) - 1)
            add(f"c2{m}{i}.{j}", [(f"c_pk({it}, {mi}, 2, {j});", "cvt")], gap_of(r2, kb, i) + LAG, dl)
    # deadlines of the element-wise chunks follow from the conversions that read them
    for kb in range(2):
        for r in range(4):
            if MODE == "fwd":
                ep[kb, r].deadline = gap_of("R1", 0, i) - MARGIN - (4 if DROP else 2)
                continue
            if "P" in MATS:
                ep[kb, r].deadline = gap_of("RP1", 0, i) - MARGIN - 1
            ed[kb, r].deadline = gap_of("R1" if MODE == "dq" else "RD1", 0, i) - MARGIN - 1
            if "P" not in MATS:
                ep[kb, r].deadline = ed[kb, r].deadline - 1

# ---- LDS fragment reads of 32-row block b32 (single-buffered registers).  A register is free from its last use by the last own
# ---- block of the previous 32-row block (+ WAR) - for the first block of a tile that use lies in the previous body for the row
# ---- fragments (stage offset 0: from gap 0 on, i.e. behind the barrier that opens the tile) and up to three slots into this body
# ---- for the transposed fragments of the output stages - and must be issued RD_AHEAD gaps ahead of its first use.
first_item = lambda b32: b32 * NB
last_item = lambda b32: b32 * NB + NB - 1


def uses(stg, pred):
    """gap positions (index k within the stage) of the MFMAs of `stg` that satisfy pred(k)"""
    n = sum(1 for (s_, _k) in LAYOUT if s_ == stg)
    return [k for k in range(n) if pred(k)]


row_stages = (("S", 0),) if MODE == "fwd" else (("S", 0), ("D", 1))
out_stages = (("O", 1),) if MODE == "fwd" else (("O", 0),) if MODE == "dq" else (("OV", 1), ("OK", 0))   # (stage, matrix whose transpose it reads)
for b32 in range(NB32):
    for stg, mat in row_stages:
        for kb in range(2):
            for pl in range(3):
                for ks in range(2):
                    # S / D index k = 12 kb + 2 product + k-step; A plane of the product = pl
                    ks_ = uses(stg, lambda k: k // 12 == kb and PROD[(k % 12) // 2][0] == pl and k % 2 == ks)
                    fu, lu = min(ks_), max(ks_)
                    use = gap_of(stg, fu, first_item(b32))
                    rel = gap_of(stg, lu, last_item(b32 - 1)) + WAR if b32 >= 1 else 0
                    assert rel <= use - RD_AHEAD, ("row fragment window", stg, kb, pl, ks, rel, use)
                    add(f"r{stg}{b32}.{kb}{pl}{ks}", [(f"rd_row({mat}, {kb}, {pl}, {ks}, {b32});", "rd")], rel, use - RD_AHEAD)
    for stg, mat in out_stages:
        for pl in range(3):
            for db in range(4):
                ks_ = uses(stg, lambda k: PROD[k // 4][0] == pl and k % 4 == db)
                fu, lu = min(ks_), max(ks_)
                for half in range(2):
                    use = gap_of(stg, fu, first_item(b32))
                    rel = max(0, gap_of(stg, lu, last_item(b32 - 1) if b32 >= 1 else last_item(NB32 - 1) - NS) + WAR)
                    assert rel <= use - RD_AHEAD, ("transposed fragment window", stg, pl, db, rel, use)
                    add(f"t{stg}{b32}.{pl}{db}{half}", [(f"rd_tr({mat}, {pl}, {db}, {half}, {b32});", "rd")], rel, use - RD_AHEAD)
    if MODE == "dkv":
        # -lse log2e / -delta of the block's 32 rows (the C operands of the first MFMA of S / dP): one float4 per 16-row block
        for kb in range(2):
            for which, stg in ((0, "S"), (1, "D")):
                use = gap_of(stg, 12 * kb, first_item(b32))
                rel = gap_of(stg, 12 * kb, last_item(b32 - 1)) + WAR if b32 >= 1 else 0
                add(f"tb{b32}.{kb}{which}", [(f"rd_tab({which}, {kb}, {b32});", "tb")], rel, use - RD_AHEAD)

if DROP and MODE == "fwd":
    # column hashes of the block's 32 keys (one uint4 per 16-row block, double-buffered by block parity: the masks of a block's last
    # item are evaluated a slot after the next block's first score product) and the hash table of the NEXT tile's 64 keys
    item_tasks = {t.name: t for t in tasks}
    for b32 in range(NB32):
        for kb in range(2):
            rel = GS if b32 == NB32 - 1 else 0        # (buffer 1's previous tenant: the previous tile's last block, masked up to slot 0)
            rd = add(f"hc{b32}.{kb}", [(f"rd_hc({b32 & 1}, {kb}, {b32});", "tb")], rel, None)
            for n_ in range(NB):
                for r in range(4):
                    item_tasks[f"ek{first_item(b32) + n_}.{kb}{r}"].after.append((rd, RD_AHEAD))
    add("hcol", [("st_hcol();", "hc")], 2 * GS, GS * (NS - 1))

# ---- staging of the next tile: LDS-DMA pieces of the pre-split tile records (2 images x 24 pieces of 1 KiB, a wavefront issues
# ---- every fourth one; dK+dV: + the piece of the rows' seeds): early in the body, so that they have landed long before the barrier
NDMA = 13 if MODE == "dkv" else 12
DMA_STEP = int(os.environ.get("GEN_DMA_STEP", "5"))           # gaps between two pieces
for j in range(NDMA):
    add(f"dma{j}", [(f"st_dma({j});", "ld")], GS // 2 + DMA_STEP * j, GS * (NS - 2) + 10)


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
        late = MODE == "fwd" and g >= G
        sched[g % G].append((call.replace("e_sum(", "e_sum_p(").replace("e_exp(", "e_exp_p(") if late else call, kind, t.name))
        t.placed.append(g)
        g += 1
    return t.deadline is None or t.placed[-1] <= t.deadline


def run(capv):
    for g in range(G):
        used[g] = 0
        sched[g] = []
    for t in tasks:
        t.placed = []
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
# the staging pieces carry their LDS base in M0, written by the first piece of a group: pieces in index order
dma_at = [t.placed[0] for t in tasks if t.name.startswith("dma")]
assert dma_at == sorted(dma_at), dma_at
emitted = [c for gap in sched for c, _k, _n in gap if c.startswith("st_dma(")]
assert emitted == [f"st_dma({j});" for j in range(NDMA)], emitted
cap = max(capv)
sys.stderr.write(f"{MODE}: {G} gaps, capacity {CAP}..{cap} cycles per gap, mean load {sum(used) / G:.1f}, max {max(used)}, "
                 f"{sum(1 for u in used if u > 8)} gaps over 8 cycles (sum of the excess {sum(max(0, u - 8) for u in used)})\n")

# ---- emit
CALL = {"S": "m_s", "D": "m_d", "O": "m_o", "OV": "m_o", "OK": "m_o"}
out = [f"// generated by tools/gen_attn6h_body.py {MODE}{' drop' if DROP else ''} - do not edit"]
for s in range(NS):
    out.append(f"// slot {s}")
    out.append(f"A6H_STAMP({s});")
    for g0, (stg, k) in enumerate(LAYOUT):
        i = s - OFFSET[stg]                 # item (negative: of the previous tile - same ring slot, same own-row block)
        it, n = i % RING, i % NB
        if stg in ("S", "D"):
            call = f"{CALL[stg]}({it}, {n}, {k});"
        elif stg.startswith("R"):
            which = 0 if MODE == "fwd" else (1 if stg in ("R1", "R2", "RD1", "RD2") else 0)
            call = f"m_r({it}, {which}, {1 if stg.endswith('1') else 2}, {k});"
        else:
            which = 0 if MODE == "fwd" else {"O": 1, "OV": 0, "OK": 1}[stg]
            call = f"m_o({it}, {n}, {which}, {k});"
        work = " ".join(c for c, _k, _n in sched[s * GS + g0] if c.split("(")[0] not in OMIT)
        out.append(f"{call} GAP_END; {work} GAP_END;".replace("  ", " "))
out.append(f"// {sum(1 for x in sched if x)} of {G} gaps carry vector work; capacity {cap} cycles per gap")
print("\n".join(out))
