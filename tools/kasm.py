#!/usr/bin/env python3
"""Instruction mix of one kernel of a translation unit (device assembly via hipcc -S).

    python tools/kasm.py quantization/mxnet_amd/csrc/fq_dwconv.hip "planes_kernel<1, true, true, 14>" [-DNAME=V ...] [--dump]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantization.mxnet_amd.csrc import build  # noqa: E402


def main():
    src, want = sys.argv[1], sys.argv[2]
    defs = [a for a in sys.argv[3:] if a.startswith("-D")]
    flags = [f for f in build.FLAGS if f not in ("-fPIC", "-shared")]
    subprocess.run([build.hipcc()] + flags + defs + ["-S", "--cuda-device-only", src, "-o", "/tmp/kasm.s"], check=True,
                   capture_output=True)
    text = open("/tmp/kasm.s").read()
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        if want not in name:
            continue
        body = m.group(2)
        if "--dump" in sys.argv:
            print(body)
        ops = collections.Counter()
        for line in body.splitlines():
            t = line.strip()
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            ops[t.split()[0]] += 1
        fam = collections.Counter()
        for k, v in ops.items():
            f = ("fp64" if "f64" in k else "dpp" if "dpp" in k else k.split("_")[0] if not k.startswith(("buffer", "global", "ds_", "flat")) else
                 k.split("_")[0] + "_" + k.split("_")[1])
            fam[f] += v
        print(name.split("(")[0])
        print("  total %d instructions: %s" % (sum(ops.values()), ", ".join("%s %d" % kv for kv in fam.most_common())))
        print("  top: " + ", ".join("%s %d" % kv for kv in ops.most_common(24)))


if __name__ == "__main__":
    main()
