#!/usr/bin/env python3
"""Registers, spills and scratch of the kernels in an object / library:  python tools/kregs.py FILE [name-substring]"""
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import isa_lint  # noqa: E402

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"


def main(path, sub=""):
    import tempfile
    for blob in isa_lint.code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as tf:
            tf.write(blob)
            tf.flush()
            r = subprocess.run([READELF, "--notes", tf.name], capture_output=True, text=True).stdout
        for b in r.split("- .agpr_count")[1:]:
            name = re.search(r"\n    \.name:\s+(\S+)", b).group(1)
            dem = subprocess.run([FILT, name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", "").replace("void ", ""))
            if sub not in dem:
                continue
            f = lambda k: re.search(r"\n    \.%s:\s+(\d+)" % k, b).group(1)
            print("%-78s vgpr %3s agpr %3s spill %3s scratch %4s lds %6s" % (dem[:78], f("vgpr_count"), re.match(r":\s+(\d+)", b).group(1),
                  f("vgpr_spill_count"), f("private_segment_fixed_size"), f("group_segment_fixed_size")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
