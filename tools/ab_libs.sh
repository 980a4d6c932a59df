#!/bin/bash
# A/B of library builds: tools/ab.py once per build (same variants), on the GPU box.
# usage: tools/ab_libs.sh "<ab.py variant args>" name1 name2 ...   (name "default" = libmi355pt.so)
set -o pipefail
ARGS=$1; shift
for n in "$@"; do
  if [ "$n" = default ]; then unset MI355PT_LIB; else export MI355PT_LIB=$PWD/pathtracer_amd/libmi355pt_$n.so; fi
  echo "== $n"
  timeout -k 10 240 python -u tools/ab.py $ARGS || exit 1
done
