#!/usr/bin/env python3
"""Generator of the tile bodies of the one-wavefront-per-SIMD bf16x6 attention kernels
(ranked-list-truncation_amd/csrc/attention6_{dkv1,dq1}_body.inc).

A tile body is a sequence of steps of six MFMAs (one six-product block each); every MFMA is followed by a "gap" that the same
wavefront fills with at most one chunk of the element-wise work (about four vector-instruction slots - what a bf16 MFMA of the
same wavefront hides: tools/micro/mfma_split.hip) and the LDS reads of the next step.  This script places the chunks: a list
scheduler, earliest deadline first, one chunk per gap, with the dependences between the accumulators, the P / dS registers, the
split fragments and the MFMAs that consume them checked at generation time.  Output: one line of calls per gap; GAP_END fences
(sched_barrier) make hipcc keep the order.

    python tools/gen_attn6_body.py dkv > ranked-list-truncation_amd/csrc/attention6_dkv1_body.inc
    python tools/gen_attn6_body.py dkv drop > ranked-list-truncation_amd/csrc/attention6_dkv1_body_drop.inc
    python tools/gen_attn6_body.py dq  > ranked-list-truncation_amd/csrc/attention6_dq1_body.inc
    python tools/gen_attn6_body.py dq drop > ranked-list-truncation_amd/csrc/attention6_dq1_body_drop.inc

Phases per tile, block b = (sub-tile b >> 1, stationary half b & 1):  X0 X1 Y0 X2 Y1 X3 Y2 Y3
  X(b): scores (4 steps) and dP (4 steps) of block b into the ONE accumulator pair;
  Y(b): the products that take the split operands of block b - dkv: dV, dK (8 steps: k-step s x {dV, dK} x d tile),
        dq: dQ (4 steps: k-step s x d tile).
Calls: mx / my (buffer, product, block, step in phase), rx / ry (buffer, read, block, step), tl / te (table reads, dkv only),
ea / eb (block, register), sp (matrix, k-step, half, part), stg (matrix, unit, part).
"""
import sys

MODE = sys.argv[1] if len(sys.argv) > 1 else "dkv"
import os
OMIT = set(os.environ.get("GEN_OMIT", "").split(","))     # diagnostic: leave out the calls of these kinds (results are wrong)
DROP = len(sys.argv) > 2 and sys.argv[2] == "drop"
AHEAD = 1                                # LDS operands of step g are read during step g - AHEAD
NB = AHEAD + 1
YLEN = 8 if MODE == "dkv" else 4
MATS = (0, 1) if MODE == "dkv" else (1,)   # split operands: P (0) and dS (1) / dS only
ORDER = [("X", 0), ("X", 1), ("Y", 0), ("X", 2), ("Y", 1), ("X", 3), ("Y", 2), ("Y", 3)]

steps = []                               # (kind, block, j)
xstart, ystart = {}, {}
for kind, b in ORDER:
    (xstart if kind == "X" else ystart)[b] = len(steps)
    for j in range(8 if kind == "X" else YLEN):
        steps.append((kind, b, j))
NS = len(steps)
NG = NS * 6
xorder = sorted(xstart, key=lambda b: xstart[b])


def next_x_after(b):
    later = [xstart[c] for c in xstart if xstart[c] > xstart[b]]
    return min(later) if later else None


def first_use(b, m, s):
    """step (in the tile) of the first MFMA that takes fragment (m, s) of block b, and of the last one"""
    if MODE == "dkv":                    # Y step j: s = j >> 2, matrix = (j >> 1) & 1, d tile = j & 1
        j0 = 4 * s + 2 * m
    else:                                # Y step j: s = j >> 1, d tile = j & 1
        j0 = 2 * s
    return ystart[b] + j0, ystart[b] + j0 + 1


class Task:
    def __init__(self, name, chunks, release, deadline):
        self.name, self.chunks, self.release, self.deadline = name, list(chunks), release, deadline
        self.done_at = None
        self.first_at = None
        self.after = []                  # (task, lag): ready at q >= task.done_at + lag

    def ready(self, q):
        if q < self.release:
            return False
        return all(t.done_at is not None and q >= t.done_at + lag for t, lag in self.after)


tasks = []
ACC_LAG = 3                              # gaps between the last MFMA of a product and the first read of its accumulator (>= 11 wait states: the asm MFMAs of the dQ kernel)
ea, eb, spl = {}, {}, {}
for b in range(4):
    x0 = xstart[b]
    nx = next_x_after(b)
    s_done = 6 * (x0 + 3) + 5
    d_done = 6 * (x0 + 7) + 5
    ea_dead = 6 * nx if nx is not None else NG              # the score accumulator is rewritten by the next X phase
    eb_dead = 6 * (nx + 4) if nx is not None else NG
    for c in range(4):
        ta = Task(f"ea{b}{c}", [f"ea({b}, {4 * c + i});" for i in range(4)], s_done + ACC_LAG, ea_dead)
        tb = Task(f"eb{b}{c}", [f"eb({b}, {4 * c + i});" for i in range(4)], d_done + ACC_LAG, eb_dead)
        tb.after.append((ta, 1))
        if c:
            ta.after.append((ea[b, c - 1], 0))
            tb.after.append((eb[b, c - 1], 0))
        ea[b, c], eb[b, c] = ta, tb
        tasks += [ta, tb]
    for s in range(2):
        for m in MATS:
            for half in range(2):
                use, _ = first_use(b, m, s)
                t = Task(f"sp{b}{m}{s}{half}", [f"sp({m}, {s}, {half}, {i});" for i in range(6)], 0, 6 * use)
                # (with dropout the mask of P is applied by the dS chunk: the split of P waits for it as well)
                t.after.append(((eb if m or DROP else ea)[b, 2 * s + half], 1))
                spl[b, m, s, half] = t
                tasks.append(t)
for k in range(1, 4):
    b, pb = xorder[k], xorder[k - 1]
    for c in range(4):
        # P / dS registers are single-buffered: block b may overwrite registers 4c..4c+3 once parts 0..2 of the split of block pb
        # have read them
        if 0 in MATS:
            ea[b, c].after.append((spl[pb, 0, c >> 1, c & 1], -2))
        else:
            ea[b, c].after.append((eb[pb, c], 1))          # dq: P lives from ea to eb of the same register only
        eb[b, c].after.append((spl[pb, 1, c >> 1, c & 1], -2))
    for s in range(2):
        for m in MATS:
            _, last = first_use(pb, m, s)
            for half in range(2):                            # fragments are single-buffered per (m, s)
                spl[b, m, s, half].release = max(spl[b, m, s, half].release, 6 * last + 5 + 1)
# staging: matrix 0 part i (registers loaded during the previous tile), then matrix 1 part i (loaded when part i of matrix 0 has
# been stored), about half a tile later; both early enough for the loads they issue
stq, std = [], []
half_tile = NG // 2 - 40
for i in range(4):
    t = Task(f"st0{i}", [f"stg(0, {i}, {k});" for k in range(6)], 0, 6 * 8)
    stq.append(t)
    tasks.append(t)
for i in range(4):
    t = Task(f"st1{i}", [f"stg(1, {i}, {k});" for k in range(6)], 0, NG * 2 // 3)
    t.after.append((stq[i], half_tile))
    std.append(t)
    tasks.append(t)

def run(cap):
    for t in tasks:
        t.done_at = t.first_at = None
    sched = [[] for _ in range(NG)]
    pending = list(tasks)
    state = {t.name: 0 for t in tasks}
    for q in range(NG):
        for _ in range(cap[q]):
            cands = [t for t in pending if t.ready(q) and (t.first_at != q or state[t.name] == 0) and state.get(("at", t.name)) != q]
            if not cands:
                break
            t = min(cands, key=lambda t: (t.deadline, tasks.index(t)))
            k = state[t.name]
            if k == 0:
                t.first_at = q
            sched[q].append(t.chunks[k])
            state[t.name] = k + 1
            state["at", t.name] = q                          # at most one chunk of a unit per gap (its parts depend on each other)
            if k + 1 == len(t.chunks):
                t.done_at = q
                pending.remove(t)
                if q >= t.deadline:
                    return None, (t, q)
    if pending:
        return None, (pending[0], NG)
    return sched, None


cap = [1] * NG
for attempt in range(40):
    sched, miss = run(cap)
    if sched is not None:
        break
    t, q = miss
    d = q - t.deadline + 1
    lo = max(0, t.deadline - 4 * d - 8)
    for x in range(lo, min(NG, t.deadline)):
        cap[x] = 2
else:
    sys.exit(f"no schedule: {miss[0].name} at {miss[1]}")
sys.stderr.write(f"{MODE}: {sum(1 for x in sched if len(x) > 1)} gaps carry two chunks\n")

pref = [[] for _ in range(NG)]
if MODE == "dq" and DROP:
    # column hashes of the keys of group (b, c) for the dS chunks, double-buffered by c & 1
    for (b, c), t in eb.items():
        lo = 0
        if c >= 2:
            lo = eb[b, c - 2].done_at + 1
        else:
            k = xorder.index(b)
            if k:
                lo = eb[xorder[k - 1], c + 2].done_at + 1
        q = max(lo, t.first_at - 9)
        if q > t.first_at - 1:           # (in the crowded tail of the tile the read may sit one gap ahead only: a short wait)
            sys.exit(f"no room for the table read of {t.name}: {q} vs first chunk at {t.first_at}")
        pref[q].append(f"te({b}, {c});")
if MODE == "dkv":
    # table reads: lse (for ea) / delta (for eb) of group (b, c), double-buffered by c & 1: at least 4 gaps before the first chunk
    # of the group, after the last chunk of the group that used the buffer before
    for kind, grp in (("tl", ea), ("te", eb)):
        for (b, c), t in grp.items():
            lo = 0
            if c >= 2:
                lo = grp[b, c - 2].done_at + 1
            else:
                k = xorder.index(b)
                if k:
                    lo = grp[xorder[k - 1], c + 2].done_at + 1
            q = max(lo, t.first_at - 9)
            if q > t.first_at - 4:
                sys.exit(f"no room for the table read of {t.name}: {q} vs first chunk at {t.first_at}")
            pref[q].append(f"{kind}({b}, {c});")


def nreads(g):
    if g >= NS:
        return 0
    kind, b, j = steps[g]
    return (4 if j < 4 else 3) if kind == "X" else 6


out = [f"// generated by tools/gen_attn6_body.py {MODE}{' drop' if DROP else ''} - do not edit"]
idle = 0
last_kind_b = None
phase = 0
for g, (kind, b, j) in enumerate(steps):
    out.append(f"// step {g}: {kind}{b}.{j}")
    out.append(f"ATTN6_STAMP({g});")
    gn = g + AHEAD
    for i in range(6):
        q = 6 * g + i
        line = f"m{kind.lower()}({g % NB}, {i}, {b}, {j}); GAP_END;"
        if kind == "X" and i == 5 and j in (3, 7):
            line += f" ACC_DONE({j >> 2});"              # the score (0) / dP (1) accumulator is complete
        if i < nreads(gn):
            k2, b2, j2 = steps[gn]
            line += f" r{k2.lower()}({gn % NB}, {i}, {b2}, {j2});"
        work = [c for c in pref[q] + sched[q] if c.split("(")[0] not in OMIT]
        line += " " + " ".join(work)
        if not sched[q]:
            idle += 1
        out.append(line.rstrip() + " GAP_END;")
out.append(f"// {NG - idle} of {NG} gaps carry a chunk")
print("\n".join(out))
