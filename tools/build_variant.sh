#!/bin/bash
# Builds the library once more with extra compiler flags, as pathtracer_amd/libmi355pt_<name>.so (development aid:
# MI355PT_LIB=<that file> makes the Python binding load it).  usage: tools/build_variant.sh <name> [-DPT_...=...]
set -e
NAME=$1; shift
cd "$(dirname "$0")/../pathtracer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall \
  -Wno-unused-function "$@" -shared -o ../libmi355pt_$NAME.so api.cpp bvh.cpp scene_io.cpp scene_pack.cpp kernels.hip lbvh.hip \
  -lz -ldl -Wl,-rpath,/opt/rocm/lib
echo built pathtracer_amd/libmi355pt_$NAME.so
