#!/bin/bash
# Builds the library as it was at a git revision, as pathtracer_amd/libmi355pt_<name>.so (A/B against the working tree).
# usage: tools/build_rev.sh <name> <rev>
set -e
NAME=$1; REV=${2:-HEAD}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" pathtracer_amd/csrc include | tar -x -C "$TMP"
cd "$TMP/pathtracer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall \
  -Wno-unused-function -shared -o "$ROOT/pathtracer_amd/libmi355pt_$NAME.so" api.cpp bvh.cpp scene_io.cpp scene_pack.cpp kernels.hip lbvh.hip \
  -lz -ldl -Wl,-rpath,/opt/rocm/lib
rm -rf "$TMP"
echo built pathtracer_amd/libmi355pt_$NAME.so from $REV
