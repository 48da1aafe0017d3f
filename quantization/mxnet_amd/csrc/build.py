"""Build libfakequant.so in-tree with plain hipcc for gfx950 (no JIT cache: the .so travels with the repo snapshot).

    python -m quantization.mxnet_amd.csrc.build [--force]

Flags that matter for parity: `-ffp-contract=off` (HIP defaults to fast contraction; a fused multiply-add would change
Winograd/EMA/interpolation results against the oracle) and NO fast-math (IEEE fp32 division and roundf decide the
integer codes).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
SRC = os.path.join(HERE, "fakequant.hip")
OUT = os.path.join(HERE, "libfakequant.so")
INCLUDE = os.path.join(ROOT, "include")
ARCH = "gfx950"


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def command(out=OUT):
    return [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
            "-fno-fast-math", "-Wall", "-Wno-unused-function", "-I", INCLUDE, SRC, "-o", out]


def up_to_date():
    if not os.path.exists(OUT):
        return False
    deps = [SRC, os.path.join(INCLUDE, "fakequant.h"), os.path.abspath(__file__)]
    return all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)


def build_library(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    cmd = command()
    if verbose:
        print("[build] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(OUT)
