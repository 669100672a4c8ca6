"""The generated kernel bodies committed under csrc/ are what their generators produce (tools/gen_attn6_body.py,
tools/gen_gemm6e_slot.py): the schedulers check every dependence of a tile body at generation time, so a body edited by hand -
or a generator changed without regenerating - must not pass."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "ranked-list-truncation_amd", "csrc")

CASES = [
    (["tools/gen_attn6_body.py", "dkv"], "attention6_dkv1_body.inc"),
    (["tools/gen_attn6_body.py", "dkv", "drop"], "attention6_dkv1_body_drop.inc"),
    (["tools/gen_attn6_body.py", "dq"], "attention6_dq1_body.inc"),
    (["tools/gen_attn6_body.py", "dq", "drop"], "attention6_dq1_body_drop.inc"),
    (["tools/gen_gemm6e_slot.py"], "gemm6e_slot.inc"),
    (["tools/gen_attn6n_body.py", "dq"], "attention6n_dq1_body.inc"),
    (["tools/gen_attn6n_body.py", "dkv"], "attention6n_dkv1_body.inc"),
    (["tools/gen_attn6n_body.py", "fwd"], "attention6n_fwd1_body.inc"),
    (["tools/gen_attn6h_body.py", "fwd"], "attention6h_fwd1_body.inc"),
    (["tools/gen_attn6h_body.py", "fwd", "drop"], "attention6h_fwd1_body_drop.inc"),
    (["tools/gen_lstm6w_body.py", "fwd"], "lstm6w_fwd_body.inc"),
    (["tools/gen_lstm6w_body.py", "fwd_xin"], "lstm6w_fwd_xin_body.inc"),
    (["tools/gen_lstm6w_body.py", "fwd_seq"], "lstm6w_fwd_seq_body.inc"),
    (["tools/gen_lstm6w_body.py", "fwd_xin_seq"], "lstm6w_fwd_xin_seq_body.inc"),
    (["tools/gen_lstm6w_body.py", "bwd"], "lstm6w_bwd_body.inc"),
    (["tools/gen_gemm6s_body.py"], "gemm6s_body.inc"),
    (["tools/gen_gemm6s_body.py", "k128"], "gemm6s_k128_body.inc"),
    (["tools/gen_gemm6s_body.py", "k128n128"], "gemm6s_k128n128_body.inc"),
]


@pytest.mark.parametrize("cmd,inc", CASES, ids=[c[1] for c in CASES])
def test_generated_body_is_current(cmd, inc):
    env = {k: v for k, v in os.environ.items() if k not in ("GEN_OMIT", "GEN_BUDGET")}
    out = subprocess.run([sys.executable] + cmd, cwd=REPO, env=env, check=True, capture_output=True, text=True).stdout
    with open(os.path.join(CSRC, inc)) as f:
        assert f.read() == out, f"{inc} is not the output of {' '.join(cmd)}"


def test_attention_bodies_cover_every_mfma_once():
    """64 (dK+dV) / 48 (dQ) steps of six MFMAs, every (step, product) exactly once, in order."""
    import re
    for inc, nstep in (("attention6_dkv1_body.inc", 64), ("attention6_dq1_body.inc", 48)):
        seen = []
        for line in open(os.path.join(CSRC, inc)):
            m = re.match(r"m([xy])\((\d), (\d), (\d), (\d)\);", line)
            if m:
                seen.append((m.group(1), int(m.group(3)), int(m.group(4)), int(m.group(5))))
        assert len(seen) == 6 * nstep
        assert [s[1] for s in seen] == [i for _ in range(nstep) for i in range(6)]
        steps = seen[::6]
        assert len(set(steps)) == nstep          # (kind, product 0, block, step in phase) distinct per step


def test_attention6n_bodies_cover_every_mfma_and_chunk_once():
    """The pipelined head-dim-16 bodies: 16 slots of 22 (dQ) / 32 (dK+dV) MFMAs, every (stage, item, index) once; every
    element-wise chunk, conversion, fragment read and staging call exactly once per tile body."""
    import re
    from collections import Counter
    for inc, per_slot, nconv in (("attention6n_dq1_body.inc", 22, 12), ("attention6n_dkv1_body.inc", 32, 24)):
        text = open(os.path.join(CSRC, inc)).read()
        mf = re.findall(r"^(m_[sdro])\(([^)]*)\); GAP_END;", text, flags=re.M)
        assert len(mf) == 16 * per_slot
        # item i and i + 4 share a ring slot and an own-row block (the residual MFMAs m_r: each call 16 / 4 times); the row and
        # output products also name the fragment buffer of their 32-row block, which alternates: 16 / 8 times
        assert {name: set(v for (nm, _a), v in Counter(mf).items() if nm == name) for name in ("m_r", "m_s", "m_d", "m_o")} == \
            {"m_r": {4}, "m_s": {2}, "m_d": {2}, "m_o": {2}}
        calls = Counter(re.findall(r"\b([a-z_]+)\(", text))
        assert calls["e_exp"] == 16 * 8 and calls["e_mul"] == 16 * 8 and calls["c_pk"] == 16 * nconv
        assert calls["rd_row"] == 4 * 12 and calls["rd_tr"] == 4 * 6 * (2 if per_slot == 32 else 1)
        assert calls["st_dma"] == (7 if per_slot == 32 else 6)       # LDS-DMA pieces of the next tile per wavefront
        # ring-indexed chunks four times per body (once per item of the ring slot), reads and staging calls once
        ring = Counter(re.findall(r"\b((?:e_exp|e_mul|c_pk)\([^)]*\))", text))
        assert set(ring.values()) == {4}
        once = Counter(re.findall(r"\b((?:rd_row|rd_tr|rd_tab|st_dma)\([^)]*\))", text))
        assert set(once.values()) == {1}


def test_attention6h_body_covers_every_mfma_and_chunk_once():
    """The pipelined head-dim-64 forward body: 8 slots of 52 MFMAs (S x24, R1 x2, R2 x2, O x24 of four different items), every
    (stage, ring slot, own block, index) the right number of times; every exp2 / row-sum / conversion chunk once per item, every
    fragment read and staging piece once per tile body, the staging pieces in index order (the first of a group writes M0)."""
    import re
    from collections import Counter
    text = open(os.path.join(CSRC, "attention6h_fwd1_body.inc")).read()
    mf = re.findall(r"^(m_[sro])\(([^)]*)\); GAP_END;", text, flags=re.M)
    assert len(mf) == 8 * 52
    by = Counter(mf)
    # items i and i + 4 share a ring slot and an own-row block: every S / O call twice, every residual call 8 / 4 = twice per ring slot x 2
    assert {n: set(v for (nm, _a), v in by.items() if nm == n) for n in ("m_s", "m_o", "m_r")} == {"m_s": {2}, "m_o": {2}, "m_r": {2}}
    assert sum(1 for (nm, _a) in mf if nm == "m_s") == 8 * 24 and sum(1 for (nm, _a) in mf if nm == "m_o") == 8 * 24
    calls = Counter(re.findall(r"\b([a-z_]+)\(", text))
    assert calls["e_exp"] + calls["e_exp_p"] == 8 * 8 and calls["e_sum"] + calls["e_sum_p"] == 8 * 8 and calls["c_pk"] == 8 * 12
    assert calls["rd_row"] == 2 * 12 and calls["rd_tr"] == 2 * 24 and calls["st_dma"] == 12
    once = Counter(re.findall(r"\b((?:rd_row|rd_tr|st_dma)\([^)]*\))", text))
    assert set(once.values()) == {1}
    assert [int(j) for j in re.findall(r"st_dma\((\d+)\)", text)] == list(range(12))


def test_attention6n_pipelined_loop_has_no_unpadded_register_moves(tmp_path):
    """The MFMAs of the pipelined bodies are asm statements: hipcc pads nothing in front of them, so the tile loop
    must not contain compiler-made register traffic into their operands - no v_accvgpr_* move and no scratch access inside
    the loop block, and exactly the generated number of MFMAs.  (A v_accvgpr_write right in front of an asm MFMA is how round 5's
    first asm build read stale operands.)"""
    import re
    import shutil
    from rlt_hip import build as B
    if shutil.which(B.HIPCC) is None and not os.path.exists(B.HIPCC):
        pytest.skip("no hipcc")
    asm = tmp_path / "a6n.s"
    src = os.path.join(CSRC, "attention6n.hip")
    subprocess.run([B.HIPCC] + B.FLAGS + B.FILE_FLAGS["attention6n.hip"] + ["-S", "--cuda-device-only", src, "-o", str(asm)], check=True,
                   capture_output=True)
    text = asm.read_text()
    for tag, nmf in (("bwd1_kernelILb0EEEv8AttnArgs", 16 * 22), ("bwd1_kernelILb1EEEv8AttnArgs", 16 * 32), ("fwd1_kernelE8AttnArgs", 16 * 16)):
        m = re.search(r"^_ZN12_GLOBAL__N_118attn6n_" + tag + r":(.*?)s_endpgm", text, flags=re.S | re.M)
        assert m, tag
        blocks = re.split(r"^\.LBB\w+:", m.group(1), flags=re.M)
        loop = max(blocks, key=lambda b: b.count("v_mfma"))
        assert loop.count("v_mfma_f32_16x16x32_bf16") == nmf
        assert "v_accvgpr" not in loop and "scratch_" not in loop


def test_attention6h_pipelined_loop_has_no_unpadded_register_moves(tmp_path):
    """The same ISA check for the head-dim-64 forward (csrc/attention6h.hip): 416 asm MFMAs in the tile loop, no v_accvgpr move, no
    scratch access, no register-to-register copy, and 12 LDS-DMA pieces with M0 written four times."""
    import re
    import shutil
    from rlt_hip import build as B
    if shutil.which(B.HIPCC) is None and not os.path.exists(B.HIPCC):
        pytest.skip("no hipcc")
    asm = tmp_path / "a6h.s"
    src = os.path.join(CSRC, "attention6h.hip")
    subprocess.run([B.HIPCC] + B.FLAGS + B.FILE_FLAGS["attention6h.hip"] + ["-S", "--cuda-device-only", src, "-o", str(asm)], check=True,
                   capture_output=True)
    text = asm.read_text()
    for tag in ("ILb0EEEv8AttnArgs", "ILb1EEEv8AttnArgs"):             # without / with dropout
        m = re.search(r"^_ZN12_GLOBAL__N_118attn6h_fwd1_kernel" + tag + r":(.*?)s_endpgm", text, flags=re.S | re.M)
        assert m, tag
        blocks = re.split(r"^\.LBB\w+:", m.group(1), flags=re.M)
        loop = max(blocks, key=lambda b: b.count("v_mfma"))
        assert loop.count("v_mfma_f32_16x16x32_bf16") == 8 * 52
        assert "v_accvgpr" not in loop and "scratch_" not in loop
        assert not re.search(r"v_mov_b32_e32 v\d+, [va]\d+", loop)          # no register-to-register copy (constants for addresses are fine)
        assert loop.count("global_load_lds_dwordx4") == 12 and len(re.findall(r"s_mov_b32 m0,", loop)) == 4
    assert re.search(r"\.vgpr_spill_count:\s+0", text[text.index("attn6h_fwd1_kernel"):] if ".vgpr_spill_count" in text else ".vgpr_spill_count: 0")


def test_lstm6w_bodies_cover_every_mfma_once():
    """Tick bodies of the whole-weights BiLSTM recurrences: every (block, k-step, product) of the chain exactly once, each element-wise
    micro-op and each LDS fragment read exactly once."""
    import re
    from collections import Counter
    for inc, nmf, macro in (("lstm6w_fwd_body.inc", 192, "MF"), ("lstm6w_fwd_xin_body.inc", 192, "MF"), ("lstm6w_fwd_seq_body.inc", 192, "MF"),
                            ("lstm6w_fwd_xin_seq_body.inc", 192, "MF"), ("lstm6w_bwd_body.inc", 192, "MB")):
        text = open(os.path.join(CSRC, inc)).read()
        mf = re.findall(r"\b" + macro + r"\((\d+), (\d+), (\d+), (\d+)\)", text)
        assert len(mf) == nmf and len(set(m[:3] for m in mf)) == nmf
        firsts = [m for m in mf if m[3] == "1"]
        if inc in ("lstm6w_fwd_body.inc", "lstm6w_fwd_seq_body.inc"):
            assert sorted(int(m[0]) for m in firsts) == list(range(8)) and all(m[1] == "0" and m[2] == "0" for m in firsts)
        elif inc == "lstm6w_bwd_body.inc":
            assert sorted(int(m[0]) for m in firsts) == [0, 1] and all(m[1] == "0" and m[2] == "0" for m in firsts)
        else:
            assert not firsts and len(re.findall(r"\bMX\(", text)) == 8       # the input projection starts the accumulators
        calls = Counter(re.findall(r"\b([A-Z][A-Z0-9]*\([^)]*\))", text))
        assert set(calls.values()) == {1}
        # every accumulator's products in the order smallest first within a k-step
        seq = {}
        for a, ks, p, _f in mf:
            seq.setdefault((a, ks), []).append(int(p))
        assert all(v == [0, 1, 2, 3, 4, 5] for v in seq.values())


def test_lstm6w_loops_have_no_compiler_register_traffic(tmp_path):
    """The MFMAs of the recurrences are asm statements (nothing is padded in front of them): their tick loops must hold no
    v_accvgpr_* move, no scratch access and no FLAT access (a FLAT access makes hipcc wait with vmcnt(0) / lgkmcnt(0) in every tick)."""
    import re
    import shutil
    from rlt_hip import build as B
    if shutil.which(B.HIPCC) is None and not os.path.exists(B.HIPCC):
        pytest.skip("no hipcc")
    asm = tmp_path / "w6.s"
    src = os.path.join(CSRC, "lstm6w.hip")
    subprocess.run([B.HIPCC] + B.FLAGS + B.FILE_FLAGS["lstm6w.hip"] + ["-S", "--cuda-device-only", src, "-o", str(asm)], check=True,
                   capture_output=True)
    text = asm.read_text()
    kernels = re.findall(r"^(_ZN12_GLOBAL__N_119bilstm6w_\w+):(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) == 6                  # forward: {pre-activations | fused input projection} x {two halves | one half}; backward x 2
    for name, body in kernels:
        assert "scratch_" not in body, name                       # nothing spills anywhere in the kernel
        blocks = re.split(r"^\.LBB\w+:", body, flags=re.M)
        ticks = [b for b in blocks if b.count("v_mfma") >= 192]                 # every block that holds tick bodies
        loops = [b for b in ticks if b.count("v_mfma_f32_16x16x32_bf16") in (192, 200, 384, 400) and "s_cbranch_scc" in b and "Loop Header" in b]   # the steady state
        assert loops, name
        for b in ticks:
            assert "v_accvgpr" not in b and "flat_" not in b, name
        for loop in loops:
            assert "vmcnt(0)" not in loop, name


def test_gemm6s_body_and_loop(tmp_path):
    """The weights-stationary K = 256 product: 384 MFMAs per block - every (half, k-step, product, column block) once, smallest product first
    per accumulator and k-step - each staging / store call once; and the block loop of every instantiation is ONE basic block without
    compiler register traffic, FLAT accesses, waterfall loops around buffer instructions or vmcnt(0)."""
    import re
    import shutil
    from collections import Counter
    for inc, KS, NCB in (("gemm6s_body.inc", 8, 4), ("gemm6s_k128_body.inc", 4, 4), ("gemm6s_k128n128_body.inc", 4, 2)):
        text = open(os.path.join(CSRC, inc)).read()
        mf = re.findall(r"\bMG\((\d), (\d), (\d), (\d), (\d)\)", text)
        assert len(mf) == 12 * NCB * KS and len(set(m[:4] for m in mf)) == 12 * NCB * KS
        assert sorted((m[0], m[3]) for m in mf if m[4] == "1") == sorted((str(h), str(c)) for h in range(2) for c in range(NCB))
        seq = {}
        for h, ks, p, cb, _f in mf:
            seq.setdefault((h, ks, cb), []).append(int(p))
        assert all(v == [0, 1, 2, 3, 4, 5] for v in seq.values())
        calls = Counter(re.findall(r"\b([A-Z][A-Z]\([^)]*\))", text))
        # (the LDS-resident l fragments of k-steps 4-7 are read once per half: twice per block)
        assert all(v == (2 if c.startswith("RL(") else 1) for c, v in calls.items())
        assert sum(1 for c in calls if c.startswith("EP(")) == 2 * NCB and sum(1 for c in calls if c.startswith("SX(")) == 7 * KS // 2
    from rlt_hip import build as B
    if shutil.which(B.HIPCC) is None and not os.path.exists(B.HIPCC):
        pytest.skip("no hipcc")
    asm = tmp_path / "g6s.s"
    subprocess.run([B.HIPCC] + B.FLAGS + B.FILE_FLAGS["gemm6s.hip"] + ["-S", "--cuda-device-only", os.path.join(CSRC, "gemm6s.hip"), "-o", str(asm)],
                   check=True, capture_output=True)
    kernels = re.findall(r"^(_ZN12_GLOBAL__N_113gemm6s_kernel\w+):(.*?)s_endpgm", asm.read_text(), flags=re.S | re.M)
    assert len(kernels) == 20                 # K = 256 / 128 x two weight layouts x four epilogues + 128-column panels (K = 128, two epilogues)
    for name, body in kernels:
        assert "scratch_" not in body, name
        blocks = re.split(r"^\.LBB\w+:", body, flags=re.M)
        loops = [b for b in blocks if b.count("v_mfma_f32_16x16x32_bf16") in (768, 384, 192)]
        assert len(loops) == 1, name                                # two blocks per iteration, one basic block
        loop = loops[0]
        assert "v_accvgpr" not in loop and "flat_" not in loop and "vmcnt(0)" not in loop and "s_cbranch_execnz" not in loop, name


def test_attention6n_forward_body_covers_every_mfma_and_chunk_once():
    """The pipelined head-dim-16 forward body: 16 slots of 16 MFMAs (score products, two residual levels, the output products); 128 exponentials
    and 128 row-sum updates per body - the chunks of a tile's last items that land in the next body under their own names."""
    import re
    from collections import Counter
    text = open(os.path.join(CSRC, "attention6n_fwd1_body.inc")).read()
    mf = re.findall(r"^(m_[sro])\(([^)]*)\); GAP_END;", text, flags=re.M)
    assert len(mf) == 16 * 16
    assert Counter(nm for nm, _a in mf) == {"m_s": 96, "m_r": 64, "m_o": 96}
    n = lambda name: len(re.findall(r"\b" + name + r"\(", text))
    assert n("e_exp") + n("e_exp_p") == 128 and n("e_sum") + n("e_sum_p") == 128 and n("e_exp_p") > 0 and n("e_sum_p") > 0
    assert n("c_pk") == 16 * 12 and n("rd_row") == 4 * 6 and n("rd_tr") == 4 * 6 and n("st_dma") == 6
    once = Counter(re.findall(r"\b((?:rd_row|rd_tr|st_dma)\([^)]*\))", text))
    assert set(once.values()) == {1}
