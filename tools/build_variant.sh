#!/bin/bash
# build_variant.sh <name> [-DNAME[=V] ...] [--amalgamate] -> build_tools/lib_<name>.so (for same-box A/B runs via
# FQ_LIB_PATH).  Objects of a variant live under csrc/build/<hash of the defines>/, so variants rebuild incrementally.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_tools
python -m quantization.mxnet_amd.csrc.build "$@" -o build_tools/lib_$name.so | tail -1
