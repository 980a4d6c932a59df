#!/bin/bash
# A/B of library builds over the three reference scenes at 1920x1080 x 256 frames.  usage: tools/ab_scenes3.sh name1 name2 ...
set -o pipefail
for s in scene1 scene2 scene3; do
  echo "#### $s"
  AB_SCENE=fixtures/$s.ron AB_F=256 AB_ROUNDS=7 tools/ab_libs.sh "x=" "$@" | grep -E "==|median" || exit 1
done
