#!/usr/bin/env python3
"""Line-level overlap of this repo's host glue with the same-named reference files (code lines only: comments, blank
lines, docstrings and licence headers stripped; a line counts when its stripped text occurs in the reference file).
Runs only where /root/reference exists (the build container); prints one row per file.

    python tools/ref_overlap.py [--ref /root/reference]"""
import argparse
import io
import os
import sys
import tokenize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [
    ("examples/simulate_quantization.py", "examples/simulate_quantization.py"),
    ("quantization/mxnet_amd/quantize/convert/convert.py", "quantize/convert/convert.py"),
    ("quantization/mxnet_amd/quantize/convert/convert_act.py", "quantize/convert/convert_act.py"),
    ("quantization/mxnet_amd/quantize/convert/convert_bn.py", "quantize/convert/convert_bn.py"),
    ("quantization/mxnet_amd/quantize/convert/convert_conv2d.py", "quantize/convert/convert_conv2d.py"),
    ("quantization/mxnet_amd/quantize/convert/convert_dense.py", "quantize/convert/convert_dense.py"),
    ("quantization/mxnet_amd/quantize/convert/ste_func.py", "quantize/convert/ste_func.py"),
    ("quantization/mxnet_amd/quantize/convert/wino_matrix.py", "quantize/convert/wino_matrix.py"),
    ("quantization/mxnet_amd/quantize/utils.py", "quantize/utils.py"),
    ("quantization/mxnet_amd/quantize/distribution_calibrate.py", "quantize/distribution_calibrate.py"),
    ("quantization/mxnet_amd/quantize/initialize/initialize.py", "quantize/initialize/initialize.py"),
    ("quantization/mxnet_amd/quantize/freeze/merge_bn.py", "quantize/freeze/merge_bn.py"),
    ("quantization/mxnet_amd/nn/quantized_conv.py", "nn/quantized_conv.py"),
]


def code_lines(path):
    src = open(path).read()
    drop = set()
    try:
        toks = list(tokenize.generate_tokens(io.StringIO(src).readline))
        prev = None
        for t in toks:
            if t.type == tokenize.COMMENT:
                pass
            if t.type == tokenize.STRING and (prev is None or prev.type in (tokenize.NEWLINE, tokenize.INDENT,
                                                                           tokenize.DEDENT, tokenize.NL)):
                for ln in range(t.start[0], t.end[0] + 1):
                    drop.add(ln)                       # docstring / bare string statement
            if t.type not in (tokenize.COMMENT, tokenize.NL):
                prev = t
    except tokenize.TokenError:
        pass
    out = []
    for i, line in enumerate(src.split("\n"), 1):
        if i in drop:
            continue
        s = line.split("#", 1)[0].strip() if "#" in line and "'#" not in line and '"#' not in line else line.strip()
        if len(s) < 4 or s in ("else:", "try:", "pass", "return", "continue", "break"):
            continue
        out.append(s)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    if not os.path.isdir(args.ref):
        print("no reference tree at %s" % args.ref)
        return 0
    worst = 0.0
    for mine, theirs in PAIRS:
        a, b = os.path.join(ROOT, mine), os.path.join(args.ref, theirs)
        if not (os.path.exists(a) and os.path.exists(b)):
            continue
        ref = code_lines(b)
        refset = set(ref)
        own = code_lines(a)
        shared_ref = sum(1 for l in ref if l in set(own))
        frac = shared_ref / max(len(ref), 1)
        worst = max(worst, frac)
        print("%-62s reference lines surviving: %3d / %3d (%4.1f %%)   own lines: %d" % (mine, shared_ref, len(ref),
                                                                                        100 * frac, len(own)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
