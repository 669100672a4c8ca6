#!/usr/bin/env python3
"""Generator of the tile body of attn6_bwd_dkv1_kernel (ranked-list-truncation_amd/csrc/attention6_dkv1_body.inc).

The body is 64 steps of six MFMAs; every MFMA is followed by a "gap" that the same wavefront fills with at most one chunk of the
element-wise work (about four vector-instruction slots, what a bf16 MFMA of the same wavefront hides: tools/micro/mfma_split.hip)
and the LDS reads of the next step.  This script places the chunks: a list scheduler, earliest deadline first, one chunk per gap,
with the dependences between the accumulators, the P / dS registers, the split fragments and the MFMAs that consume them checked
at generation time.  Output: one line of calls per gap, fenced by sched_barrier so that hipcc keeps the order.

    python tools/gen_dkv1_body.py > ranked-list-truncation_amd/csrc/attention6_dkv1_body.inc
"""
import sys

AHEAD = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # LDS operands of step g are read during step g - AHEAD

PH_KIND = [0, 0, 1, 0, 1, 0, 1, 1]      # 0: X (S, dP), 1: Y (dV, dK)
PH_BLK = [0, 1, 0, 2, 1, 3, 2, 3]
NG = 64 * 6

xstart = {}                              # block -> first step of its X phase
ystart = {}
for p in range(8):
    (ystart if PH_KIND[p] else xstart)[PH_BLK[p]] = 8 * p
xorder = sorted(xstart, key=lambda b: xstart[b])


def next_x_after(b):
    later = [xstart[c] for c in xstart if xstart[c] > xstart[b]]
    return min(later) if later else None


class Task:
    def __init__(self, name, chunks, release, deadline):
        self.name, self.chunks, self.release, self.deadline = name, list(chunks), release, deadline
        self.done_at = None              # gap of the last chunk
        self.first_at = None
        self.after = []                  # (task, lag): release >= task.done_at + lag
        self.after_first = []            # (task, lag): release >= task.first_at + lag

    def ready(self, q):
        if q < self.release:
            return False
        for t, lag in self.after:
            if t.done_at is None or q < t.done_at + lag:
                return False
        for t, lag in self.after_first:
            if t.first_at is None or q < t.first_at + lag:
                return False
        return True


tasks = []
ACC_LAG = 2                              # gaps between the last MFMA of a product and the first read of its accumulator

ea, eb, spl = {}, {}, {}
for b in range(4):
    x0 = xstart[b]
    nx = next_x_after(b)
    s_done = 6 * (x0 + 3) + 5            # gap of the last S MFMA
    d_done = 6 * (x0 + 7) + 5
    ea_dead = 6 * nx if nx is not None else NG          # sc rewritten by the first MFMA of the next X phase
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
    # split units: (m, s, half); consumed by Y(b) steps 4 s + 2 m (+1)
    for s in range(2):
        for m in range(2):
            for half in range(2):
                use = 6 * (ystart[b] + 4 * s + 2 * m)           # gap of the first MFMA that takes the fragment
                t = Task(f"sp{b}{m}{s}{half}", [f"sp({m}, {s}, {half}, {i});" for i in range(6)], 0, use)
                t.after.append(((eb if m else ea)[b, 2 * s + half], 1))
                spl[b, m, s, half] = t
                tasks.append(t)
# P / dS registers are single-buffered: the element-wise unit of block b + 1 (in X order) may overwrite registers 4c..4c+3 only
# after the split of block b has read them (part 0 of the unit)
for k in range(1, 4):
    b, pb = xorder[k], xorder[k - 1]
    for c in range(4):
        ea[b, c].after.append((spl[pb, 0, c >> 1, c & 1], -2))      # its inputs are read by parts 0..2 of 0..5
        eb[b, c].after.append((spl[pb, 1, c >> 1, c & 1], -2))
# fragments are single-buffered per (m, s): the split of block b + 1 may write fr[m][s] only after Y(b)'s last MFMA that reads it
for k in range(1, 4):
    b, pb = xorder[k], xorder[k - 1]
    for s in range(2):
        for m in range(2):
            last_use = 6 * (ystart[pb] + 4 * s + 2 * m + 1) + 5
            for half in range(2):
                spl[b, m, s, half].release = max(spl[b, m, s, half].release, last_use + 1)
# staging: Q part i (registers loaded during the previous tile), then dO part i (loaded when Q part i has been stored), 192 gaps later
stq, std = [], []
for i in range(4):
    t = Task(f"stq{i}", [f"stg(0, {i}, {k});" for k in range(6)], 0, 6 * 8)
    stq.append(t)
    tasks.append(t)
for i in range(4):
    t = Task(f"std{i}", [f"stg(1, {i}, {k});" for k in range(6)], 0, 6 * 42)       # early enough for the loads it issues (Q of the tile after next)
    t.after.append((stq[i], 150))                                                    # ~25 steps: the latency of the load issued by stq
    std.append(t)
    tasks.append(t)

# table prefetches: issued in the gap of the first chunk of the previous group of the same kind (one step ahead at least)
sched = [[] for _ in range(NG)]
pending = list(tasks)
state = {t.name: 0 for t in tasks}
for q in range(NG):
    cands = [t for t in pending if t.ready(q)]
    if not cands:
        continue
    t = min(cands, key=lambda t: (t.deadline, tasks.index(t)))
    k = state[t.name]
    if k == 0:
        t.first_at = q
    sched[q].append(t.chunks[k])
    state[t.name] = k + 1
    if k + 1 == len(t.chunks):
        t.done_at = q
        pending.remove(t)
        if q >= t.deadline:
            sys.exit(f"deadline missed: {t.name} done at gap {q}, deadline {t.deadline}")
if pending:
    sys.exit("unscheduled: " + " ".join(t.name for t in pending))

# table reads: lse (for ea) / delta (for eb) of group (b, c), double-buffered by c & 1: at least 6 gaps before the first chunk of
# the group, after the last chunk of group c - 2 of the same kind
pref = [[] for _ in range(NG)]
for kind, grp in (("tl", ea), ("te", eb)):
    for (b, c), t in grp.items():
        lo = 0
        if c >= 2:
            lo = grp[b, c - 2].done_at + 1
        else:
            # the buffer was last used by group c + 2 of the previous block in X order
            k = xorder.index(b)
            if k:
                lo = grp[xorder[k - 1], c + 2].done_at + 1
        q = max(lo, t.first_at - 9)
        if q > t.first_at - 4:
            sys.exit(f"no room for the table read of {t.name}: {q} vs first chunk at {t.first_at}")
        pref[q].append(f"{kind}({b}, {c});")

out = []
out.append("// generated by tools/gen_dkv1_body.py - do not edit")
idle = 0
for g in range(64):
    p, j = g >> 3, g & 7
    out.append(f"// step {g}: {'XY'[PH_KIND[p]]}{PH_BLK[p]}.{j}")
    if j == 0:
        out.append(f"DKV1_STAMP({p});")
    gn = g + AHEAD
    nrd = 0 if gn > 63 else (6 if PH_KIND[gn >> 3] else (4 if (gn & 7) < 4 else 3))
    for i in range(6):
        q = 6 * g + i
        line = f"mf({g}, {i}); GAP_END;"
        if i < nrd:
            line += f" rd({gn}, {i});"
        line += " " + " ".join(pref[q] + sched[q])
        if not sched[q]:
            idle += 1
        out.append(line.rstrip() + " GAP_END;")
out.append(f"// {NG - idle} of {NG} gaps carry a chunk")
print("\n".join(out))
