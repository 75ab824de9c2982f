#!/bin/bash
# A/B of whole libraries over the three poses: scripts/r04/ab_poses.sh <tag> <lib> ...; per pose the frame rate with three frames in flight and the launch alone.
set -o pipefail
TAG=$1; shift
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/ab_poses_$TAG.txt; : > $OUT
for rep in 1 2; do
  for lib in "$@"; do
    BLOK_HIP_LIB=$PWD/$lib python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-paths 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
p = d['config']['poses']
print('$lib', ' '.join('%s: %.4f in flight, %.4f alone;' % (k, v['ms_per_frame'], v['ms_per_frame_alone']) for k, v in p.items()))" | tee -a $OUT || exit 1
  done
done
