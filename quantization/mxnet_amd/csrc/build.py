"""Build libfakequant.so in-tree with plain hipcc for gfx950 (no JIT cache: the .so travels with the repo snapshot).

    python -m quantization.mxnet_amd.csrc.build [--force] [--prune] [--dev] [--amalgamate] [-DNAME[=V] ...] [--only UNIT[,UNIT]] [-o OUT]

`--dev` also compiles the forms that were measured and shelved (DEV_UNITS: the pipe form of the small-plane pointwise layers) into
`libfakequant_dev.so`; the shipped library does not carry them.

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


# per-unit extras.  fq_pwdw: the SLP vectoriser packs the depthwise chains into v_pk_fma_f32 and pays two v_mov per pair for it
# (a packed fp32 instruction costs about 1.7 plain ones on gfx950, profiles/r5_pk_probe.txt): 72 instead of 48 instructions
UNIT_FLAGS = {"fq_pwdw": ["-fno-slp-vectorize"]}


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


# forms that were built, measured and lost: kept as evidence with their parity cases, compiled by `--dev` only
# (-DFQ_DEV_FORMS tells fq_pwconv.hip to dispatch to them; fq_build_has("pipe") tells the tests)
DEV_UNITS = {"fq_pw_pipe"}


def sources(dev=False):
    return [s for s in sorted(glob.glob(os.path.join(HERE, "fq_*.hip")))
            if dev or os.path.basename(s)[:-4] not in DEV_UNITS]


def headers():
    return sorted(glob.glob(os.path.join(HERE, "*.h"))) + [os.path.join(INCLUDE, "fakequant.h"),
                                                           os.path.abspath(__file__)]


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def source_id(defines=()):
    """sha1 over every source, every header, this script's flags and the defines: the identity of what a library SHOULD be
    built from.  It is compiled into the library (fq_build_id(), -DFQ_BUILD_ID on fq_core.hip), so whether a built file
    matches the tree is a question of content, not of modification times (a stale-but-newer .so is not "up to date")."""
    h = hashlib.sha1()
    for f in sources("-DFQ_DEV_FORMS" in defines) + [x for x in headers() if not x.endswith(".py")]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS[:-4] + list(defines) + [repr(sorted(UNIT_FLAGS.items()))]).encode())        # (the include paths are machine-specific)
    return h.hexdigest()


_ID_MARK = b"FQ_BUILD_ID="


def built_id(out=OUT):
    """The build id embedded in a library file (None when absent): read from the bytes, nothing is loaded."""
    try:
        with open(out, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    i = blob.find(_ID_MARK)
    return blob[i + len(_ID_MARK):i + len(_ID_MARK) + 40].decode("ascii", "replace") if i >= 0 else None


def up_to_date(out=OUT, defines=()):
    return built_id(out) == source_id(defines)


def prune(verbose=True):
    """Delete what no longer belongs to the tree but would still travel to the GPU box with it: objects of units whose
    source is gone, compiler temporaries, and variant directories whose objects are older than the sources they were
    built from (tools/build_variant.sh rebuilds those anyway)."""
    import shutil
    units = {os.path.basename(x)[:-4] for x in sources()}
    newest = max(os.path.getmtime(f) for f in sources() + headers())
    removed = []
    for d in sorted(glob.glob(os.path.join(OBJ_DIR, "*"))):
        if not os.path.isdir(d):
            continue
        tag = os.path.basename(d)
        for f in sorted(os.listdir(d)):
            unit = f.split(".")[0]
            stale = (f.endswith(".o") and unit not in units and unit != "fq_all") or (".o." in f and not f.endswith(".o.id"))
            if stale:
                os.remove(os.path.join(d, f))
                removed.append(os.path.join(tag, f))
        if tag != "default":
            objs = glob.glob(os.path.join(d, "*.o"))
            if not objs or all(os.path.getmtime(o) < newest for o in objs):
                shutil.rmtree(d)
                removed.append(tag + "/")
    if verbose and removed:
        print("[build] pruned " + " ".join(removed), flush=True)
    return removed


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
    if not defines:
        prune(verbose)
    tag = hashlib.sha1(" ".join(defines).encode()).hexdigest()[:8] if defines else "default"
    objdir = os.path.join(OBJ_DIR, tag)
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    hdrs = headers()
    bid = source_id(defines)
    idflag = '-DFQ_BUILD_ID="%s"' % bid
    if amalgamate:
        unit = os.path.join(objdir, "fq_all.hip")
        with open(unit, "w") as f:
            for s in sources("-DFQ_DEV_FORMS" in defines):
                f.write('#include "%s"\n' % os.path.basename(s))
        _run([cc] + FLAGS + defines + [idflag, "-shared", unit, "-o", out], verbose)
        return out
    todo, objs = [], []
    for s in sources("-DFQ_DEV_FORMS" in defines):
        unit = os.path.basename(s)[:-4]
        mine = not only or unit in only or unit == "fq_core"
        o = os.path.join(objdir if mine else os.path.join(OBJ_DIR, "default"), unit + ".o")
        objs.append(o)
        extra, fresh = [], _newer(o, [s] + hdrs)
        if unit == "fq_core":                    # carries the build id: rebuilt whenever ANY source changed
            extra = [idflag]
            idfile = o + ".id"
            fresh = fresh and os.path.exists(idfile) and open(idfile).read() == bid
        if mine and (force or not fresh):
            udefs = defines if (not only or unit in only) else []
            todo.append(([cc] + FLAGS + UNIT_FLAGS.get(unit, []) + udefs + extra + ["-c", s, "-o", o],
                         o + ".id" if unit == "fq_core" else None))
    jobs = jobs or min(len(todo) or 1, os.cpu_count() or 4)

    def compile_one(job):
        cmd, idfile = job
        _run(cmd, verbose)
        if idfile:
            with open(idfile, "w") as f:
                f.write(bid)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, todo))
    _run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", out], verbose)
    if built_id(out) != bid:
        raise RuntimeError("%s does not carry the build id of the sources it was just linked from" % out)
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    out = OUT
    if "-o" in argv:
        out = os.path.abspath(argv[argv.index("-o") + 1])
    only = argv[argv.index("--only") + 1].split(",") if "--only" in argv else ()
    if "--prune" in argv:
        prune()
    if "--dev" in argv:                                  # the shelved forms too (a library of its own unless -o names the default)
        argv = argv + ["-DFQ_DEV_FORMS"]
        if "-o" not in argv:
            out = os.path.join(HERE, "libfakequant_dev.so")
    build_library(force="--force" in argv, defines=[a for a in argv if a.startswith("-D")], out=out,
                  amalgamate="--amalgamate" in argv, only=only)
    print(out)
