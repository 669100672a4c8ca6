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
]


@pytest.mark.parametrize("cmd,inc", CASES, ids=[c[1] for c in CASES])
def test_generated_body_is_current(cmd, inc):
    env = {k: v for k, v in os.environ.items() if k != "GEN_OMIT"}
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
