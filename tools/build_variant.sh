#!/bin/bash
# build_variant.sh <name> [extra hipcc flags...] -> build_tools/lib_<name>.so (for same-box A/B runs via FQ_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_tools
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function \
  -I include "$@" quantization/mxnet_amd/csrc/fakequant.hip -o build_tools/lib_$name.so
echo build_tools/lib_$name.so
