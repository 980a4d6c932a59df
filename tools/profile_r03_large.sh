#!/bin/bash
# Counter profiles of the large-scene configurations (run via gpurun): 1 M spheres x 64 frames, 10 M x 32,
# and the 10 M-sphere plane seen from above (every tile's camera rays walk, the frame touches the whole tree).
set -o pipefail
tools/profile_cfg.sh r03_c4_1M_4k 1000000 3840 2160 64 3 || exit 1
tools/profile_cfg.sh r03_c5_10M_4k 10000000 3840 2160 32 3 || exit 1
RUN_CFG_CAMERA=0,5000,0,0,-89.5,20 tools/profile_cfg.sh r03_c5_10M_topdown 10000000 3840 2160 16 3 || exit 1
echo all done
