#!/bin/bash
# Queued steps (bench.py's timed loop) per scene and library build, on the GPU box.
# usage: tools/queue_ab.sh name1 name2 ...   (name "default" = libmi355pt.so; others from tools/build_rev.sh / build_variant.sh)
set -o pipefail
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = default ]; then unset MI355PT_LIB; else export MI355PT_LIB=$PWD/pathtracer_amd/libmi355pt_$n.so; fi
  for s in scene1 scene2 scene3; do
    timeout -k 10 200 python bench.py --scene fixtures/$s.ron --steps 8 --warmup 2 --no-cpu-baseline --no-rmse --no-configs 2>/dev/null \
      | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print('$n $s ms_per_step %.3f kernel %.3f accumulation %.3f' % (j['ms_per_step'], j['kernel_ms_per_launch'], j['resolve_ms_per_launch']))" || exit 1
  done
done
done
