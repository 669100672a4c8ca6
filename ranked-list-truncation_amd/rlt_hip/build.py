"""Build librlt_hip.so (gfx950) in-tree with hipcc.  No GPU is needed to build."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB = os.path.join(CSRC, "librlt_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]   # no implicit fma contraction: keep fp32 op order as written
# per-file extras.  attention6.hip: no SLP packing - v_pk_add_f32 / v_pk_mul_f32 do not run beside a partner wavefront's bf16 MFMAs
# (tools/micro/mfma_valu_overlap.hip), the unpacked forms do; the exact-fp32 files keep the packing (there every vector instruction
# adds to the MFMA time, and a packed one does two lanes' work)
FILE_FLAGS = {"attention6.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(PKG_DIR), "include", "rlt_hip.h"))
    objs = []
    jobs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
