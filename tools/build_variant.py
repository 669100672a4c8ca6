#!/usr/bin/env python3
"""Kernel A/B experiments: build librlt_hip.so again with extra -D flags for ONE source file, next to the product build.

    python tools/build_variant.py NAME attention3.hip -DRLT_EXP_PIPE2 [-D...]

-> ranked-list-truncation_amd/csrc/variants/librlt_NAME.so (travels to the GPU box; select it with RLT_HIP_LIB=<path>,
tools/bench_kernels.py prints kernel times).  The product library is untouched."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))
from rlt_hip import build as B  # noqa: E402


def main():
    name, src = sys.argv[1], sys.argv[2]
    flags = sys.argv[3:]
    B.build(verbose=False)                                   # product objects up to date
    out_dir = os.path.join(B.CSRC, "variants")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, f"{name}_{src[:-4]}.o")
    subprocess.run([B.HIPCC] + B.FLAGS + B.FILE_FLAGS.get(src, []) + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
    objs = [o for o in (s[:-4] + ".o" for s in B.sources()) if os.path.basename(o) != src[:-4] + ".o"] + [obj]
    lib = os.path.join(out_dir, f"librlt_{name}.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    os.remove(obj)
    print(lib)


if __name__ == "__main__":
    main()
