#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kres.py quantization/mxnet_amd/csrc/fq_pw_split.hip [-DNAME=V ...] [--filter split]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantization.mxnet_amd.csrc import build  # noqa: E402


def main():
    src = sys.argv[1]
    defs = [a for a in sys.argv[2:] if a.startswith("-D")]
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
    cmd = [build.hipcc()] + build.FLAGS + defs + ["-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"remark: +(.*?) \[-Rpass", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        if flt in r["name"]:
            print("%-60s sgpr %3s vgpr %3s agpr %3s scratch %4s occ %s lds %s" % (
                r["name"][-60:], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
                r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()
