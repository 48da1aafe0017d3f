#!/usr/bin/env python3
"""ISA lint of the built library for the gfx950 store-data hazard found in round 3 (profiles/r3_dw_flat_race.txt):

    buffer_store_dwordx4 v[a:a+3], voff, s[..], sN offen        <- wider than 8 bytes, SGPR soffset
    v_cndmask_b32 v(a), ...                                     <- a VALU write of a data register within 2 wait states

LLVM's hazard recogniser adds the wait states for wide stores only when soffset is NOT a register
(GCNHazardRecognizer::createsVALUHazard), so nothing protects this form; on MI355X the VALU write then overtakes the
store's data read under load.  The tool pulls every gfx950 code object out of the shared library (clang offload bundles in
.hip_fatbin), disassembles it with llvm-objdump and reports each wide buffer / global / flat / scratch store that is followed,
within WAIT_STATES issue slots, by a vector instruction writing one of its data registers.

    python tools/isa_lint.py [path/to/lib.so]        exit status 1 if anything is found
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WAIT_STATES = 2            # what LLVM itself applies on gfx940+ where it does see the hazard


def code_objects(path, arch="gfx950"):
    data = open(path, "rb").read()
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        (count,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if arch in triple and size:
                yield data[base + off:base + off + size]


def vregs(tok):
    """'v[4:7]' -> {4,5,6,7}; 'v8' -> {8}; anything else -> empty."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


WIDE = re.compile(r"^(buffer_store_(dwordx[34]|format_xyzw?)|global_store_dwordx[34]|flat_store_dwordx[34]|"
                  r"scratch_store_dwordx[34])\b")


def nops(ins, ops):
    if ins == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def lint_text(text):
    findings, kernel = [], None
    lines = []
    for raw in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", raw.strip())
        if m:
            kernel = m.group(1)
            continue
        body = raw.split("//")[0].strip()
        if not body or body.startswith("."):
            continue
        parts = body.replace(",", " ").split()
        lines.append((kernel, parts[0], parts[1:], body))
    for i, (kernel, ins, ops, body) in enumerate(lines):
        if not WIDE.match(ins) or not ops:
            continue
        data = vregs(ops[0]) if ins.startswith("buffer") else (vregs(ops[1]) if len(ops) > 1 else set())
        if not data:
            continue
        slots, j = 0, i + 1
        while j < len(lines) and slots < WAIT_STATES and lines[j][0] == kernel:
            _, ins2, ops2, body2 = lines[j]
            if ins2.startswith("v_") and not ins2.startswith("v_cmp") and ops2 and vregs(ops2[0]) & data:
                findings.append((kernel, body, body2, slots))
                break
            if ins2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
            slots += nops(ins2, ops2)
            j += 1
    return findings


def lint_handshake(text, kernel_substr="pwconv_rows_kernelILb1"):
    """The "last workgroup" hand-shake of the classifier + counters kernel (csrc/fq_pw_rows.hip): its cross-workgroup accesses
    are agent-scope atomics compiled to sc1 stores / loads, and the release side is an `s_waitcnt vmcnt(0)` directly in front
    of the barrier that precedes the counter's atomic add (a workgroup-scope fence alone compiles to nothing here).  Returns a
    list of problems (empty = fine; None = the kernel is not in this code object)."""
    m = re.search(r"^[0-9a-f]+ <([^>]*%s[^>]*)>:$" % re.escape(kernel_substr), text, re.M)
    if not m:
        return None
    seg = text[m.end():]
    nxt = re.search(r"^[0-9a-f]+ <", seg, re.M)
    seg = seg[:nxt.start()] if nxt else seg
    ins = [l.split("//")[0].strip() for l in seg.splitlines() if l.strip() and not l.strip().startswith(".")]
    problems = []
    if not any(re.match(r"global_store_dwordx2 .* sc1", i) for i in ins):
        problems.append("no sc1 (agent-scope, written through) 64-bit key store")
    if not any(re.match(r"global_load_dwordx2 .* sc1", i) for i in ins):
        problems.append("no sc1 (past the vector cache and L2) 64-bit key load")
    # the release side: an `s_waitcnt vmcnt(0)` of the short hand-shake's own (not the one inside the textbook path's
    # buffer_wbl2 / buffer_inv pair) with nothing but branches, waits and cache maintenance between it and the barrier
    ok = False
    harmless = ("s_nop", "s_cbranch", "s_branch", "s_waitcnt", "buffer_wbl2", "buffer_inv")
    for k, i in enumerate(ins):
        if not re.match(r"s_waitcnt .*vmcnt\(0\)", i) or (k > 0 and ins[k - 1].startswith("buffer_wbl2")):
            continue
        j = k + 1
        while j < len(ins) and j <= k + 8 and ins[j].startswith(harmless):
            j += 1
        if j < len(ins) and ins[j].startswith("s_barrier"):
            ok = True
    if not ok:
        problems.append("no `s_waitcnt vmcnt(0)` of its own in front of the release barrier: the key stores are not waited for")
    return problems


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "quantization", "mxnet_amd", "csrc", "libfakequant.so")
    total, stores, handshake = [], 0, None
    for k, blob in enumerate(code_objects(lib)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        stores += len(re.findall(r"^\s*(buffer_store_dwordx[34]|global_store_dwordx[34])", text, re.M))
        total += lint_text(text)
        hs = lint_handshake(text)
        if hs is not None:
            handshake = hs
    for kernel, st, wr, slots in total:
        name = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip()
        print("HAZARD in %s\n    %s\n    %s   (%d wait state(s) after the store)" % (name[:150], st, wr, slots))
    print("isa_lint: %d wide stores checked, %d unprotected VALU writes of store data" % (stores, len(total)))
    if handshake is None:
        print("isa_lint: pwconv_rows_kernel<true> not found: its hand-shake was NOT checked")
    else:
        for p in handshake:
            print("HANDSHAKE pwconv_rows_kernel<true>: " + p)
        print("isa_lint: last-workgroup hand-shake of pwconv_rows_kernel<true>: %s" % ("ok" if not handshake else "BROKEN"))
    sys.exit(1 if total or handshake or handshake is None else 0)


if __name__ == "__main__":
    main()
