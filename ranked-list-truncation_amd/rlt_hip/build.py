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
# attention6n.hip, -amdgpu-mfma-vgpr-form: its one-wavefront kernels own 512 registers, where hipcc would select the AGPR form of every
# MFMA and move each score register to a VGPR and back around the element-wise work (828 v_accvgpr moves per tile body of the dQ
# kernel); with the VGPR form the accumulators the vector ALU touches stay where it can reach them
# attention6h.hip: the same form at head dim 64, the same flags
# lstm6w.hip, no SLP packing: the split residuals of values that sit in different registers would be packed by first MOVING them into pairs
FILE_FLAGS = {"attention6.hip": ["-fno-slp-vectorize"], "attention6n.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"],
              "attention6h.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"],
              "lstm6w.hip": ["-fno-slp-vectorize"], "gemm6s.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]      # .inc: generated kernel bodies
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


SAN_LIB = os.path.join(CSRC, "librlt_hip_asan.so")
SAN_FLAGS = ["-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--offload-host-only", "-fsanitize=address,undefined",
             "-fno-omit-frame-pointer", "-shared-libsan", "-DRLT_BOUNDS_CHECK", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]


def sanitizer_runtime():
    """clang's shared AddressSanitizer runtime: a python process that dlopens the sanitized library needs it preloaded."""
    res = subprocess.run([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = res.stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def build_sanitized(verbose=True):
    """HOST-ONLY AddressSanitizer + UndefinedBehaviorSanitizer build of the whole library (VERDICT r03
    item 9): `--offload-host-only` compiles argument checking, workspace layout arithmetic (csrc/path.hip with
    -DRLT_BOUNDS_CHECK), split-K planning and launch set-up, no device code - kernels cannot launch from it; it exists to run
    the no-GPU ABI tests under the sanitizers on the CPU box (tests/test_sanitized_abi.py).  GPU ASan is not available on this
    pool."""
    srcs = sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]      # .inc: generated kernel bodies
    headers.append(os.path.join(os.path.dirname(PKG_DIR), "include", "rlt_hip.h"))
    if _stale(SAN_LIB, srcs + headers + [os.path.abspath(__file__)]):
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError("hipcc (sanitized host build) failed:\n" + res.stdout + res.stderr)
            return res.stdout
        objs = [src[:-4] + ".asan.o" for src in srcs]
        with ThreadPoolExecutor(max_workers=6) as ex:
            list(ex.map(lambda so: run([HIPCC] + SAN_FLAGS + ["-c", so[0], "-o", so[1]]), zip(srcs, objs)))
        # a host-only compile leaves one `__hip_fatbin_<hash>` per translation unit undefined (the device link step would
        # have supplied the code objects): give each an EMPTY clang offload bundle, so that the library loads and registers
        # "no kernels" - which is what a host-only build has
        undefined = sorted({l.split()[-1] for o in objs for l in run(["nm", "-u", o]).splitlines() if "__hip_fatbin_" in l})
        stub = SAN_LIB + ".fatbin_stubs.c"
        with open(stub, "w") as f:
            f.write("/* generated by rlt_hip/build.py: empty offload bundles for the host-only sanitizer build */\n")
            for sym in undefined:
                f.write('__attribute__((aligned(4096))) const char %s[32] = "__CLANG_OFFLOAD_BUNDLE__";\n' % sym)
        stub_o = stub[:-2] + ".o"
        run(["gcc", "-c", "-fPIC", stub, "-o", stub_o])
        run([HIPCC, "--offload-arch=gfx950", "--offload-host-only", "-fsanitize=address,undefined", "-shared-libsan", "-shared", "-fPIC",
             "-o", SAN_LIB] + objs + [stub_o])
        for f in objs + [stub, stub_o]:
            os.remove(f)
    return SAN_LIB


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        print(build_sanitized())
    else:
        print(build(force="--force" in sys.argv))
