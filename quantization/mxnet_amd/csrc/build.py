"""Build libfakequant.so in-tree with plain hipcc for gfx950 (no JIT cache: the .so travels with the repo snapshot).

    python -m quantization.mxnet_amd.csrc.build [--force] [--amalgamate] [-DNAME[=V] ...] [--only UNIT[,UNIT]] [-o OUT]

Every `fq_*.hip` translation unit is compiled to an object under `csrc/build/` (in parallel, only when it or a header
changed) and the objects are linked into one shared library.  `--amalgamate` compiles all units as ONE translation unit
instead (used by the trace build `-DFQ_PW_TRACE`, whose `__device__` debug symbols must exist once).  `--only fq_pw_sample`
(tuning variants): the defines apply to the named units only, every other object is the default build's - a variant then
costs one compilation instead of thirteen.

Flags that matter for parity: `-ffp-contract=off` (HIP defaults to fast contraction; a fused multiply-add would change
Winograd/EMA/interpolation results against the oracle) and NO fast-math (IEEE fp32 division and roundf decide the
integer codes).
"""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
OUT = os.path.join(HERE, "libfakequant.so")
OBJ_DIR = os.path.join(HERE, "build")
INCLUDE = os.path.join(ROOT, "include")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
         "-Wno-unused-function", "-I", INCLUDE, "-I", HERE]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(glob.glob(os.path.join(HERE, "fq_*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(HERE, "*.h"))) + [os.path.join(INCLUDE, "fakequant.h"),
                                                           os.path.abspath(__file__)]


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def up_to_date(out=OUT):
    return _newer(out, sources() + headers())


def _run(cmd, verbose):
    if verbose:
        print("[build] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_library(force=False, verbose=True, defines=(), out=OUT, amalgamate=False, jobs=None, only=()):
    """Returns the path of the library.  `defines` (e.g. ["-DFQ_PW_TRACE"]) select a separate object directory; with
    `only` (unit names without extension) they apply to those units and the rest is taken from the default objects."""
    defines = list(defines)
    only = set(only)
    if only:
        build_library(verbose=verbose)                       # the default objects must exist and be current
    if not force and not defines and out == OUT and up_to_date(out):
        return out
    tag = hashlib.sha1(" ".join(defines).encode()).hexdigest()[:8] if defines else "default"
    objdir = os.path.join(OBJ_DIR, tag)
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    hdrs = headers()
    if amalgamate:
        unit = os.path.join(objdir, "fq_all.hip")
        with open(unit, "w") as f:
            for s in sources():
                f.write('#include "%s"\n' % os.path.basename(s))
        _run([cc] + FLAGS + defines + ["-shared", unit, "-o", out], verbose)
        return out
    todo, objs = [], []
    for s in sources():
        unit = os.path.basename(s)[:-4]
        mine = not only or unit in only
        o = os.path.join(objdir if mine else os.path.join(OBJ_DIR, "default"), unit + ".o")
        objs.append(o)
        if mine and (force or not _newer(o, [s] + hdrs)):
            todo.append([cc] + FLAGS + defines + ["-c", s, "-o", o])
    jobs = jobs or min(len(todo) or 1, os.cpu_count() or 4)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(lambda c: _run(c, verbose), todo))
    _run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", out], verbose)
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    out = OUT
    if "-o" in argv:
        out = os.path.abspath(argv[argv.index("-o") + 1])
    only = argv[argv.index("--only") + 1].split(",") if "--only" in argv else ()
    build_library(force="--force" in argv, defines=[a for a in argv if a.startswith("-D")], out=out,
                  amalgamate="--amalgamate" in argv, only=only)
    print(out)
