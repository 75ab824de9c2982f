#!/bin/bash
# A/B on the GPU box of whole libraries: scripts/r04/ab_libs.sh <tag> <lib> [<lib> ...]; the default bench and the driver's arguments, two
# repetitions each, alternating; optionally the path kernel (PATHS=1: 64 spp, poses A and B).
set -o pipefail
TAG=$1; shift
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/ab_$TAG.txt; : > $OUT
for rep in 1 2; do
  for lib in "$@"; do
    for a in "--steps 200 --warmup 10" ${ONLY_LONG:+__skip__} "--steps 20 --warmup 5"; do
      [ "$a" = "__skip__" ] && break
      BLOK_HIP_LIB=$PWD/$lib python3 bench.py $a --no-cpu-baseline --no-paths --no-poses 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$lib', '$a', 'Mrays/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'alone %.4f' % d['config']['kernel_ms_alone'], 'moving %.4f' % d['config']['kernel_ms_alone_moving'])" | tee -a $OUT || exit 1
    done
  done
done
if [ -n "$PATHS" ]; then
  for lib in "$@"; do
    echo "== $lib" | tee -a $OUT
    BLOK_HIP_LIB=$PWD/$lib python3 scripts/r03/paths_ab.py 64 2 2>&1 | grep -v amdgpu.ids | head -${PATH_POSES:-2} | tee -a $OUT
  done
fi
