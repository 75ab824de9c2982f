#!/bin/bash
# Beam tiles of 16 / 32 / 64 pixels on the round's final searches: ms per frame, three in flight / alone, poses A B C.
set -o pipefail
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/beam_tile_sweep.txt; : > $OUT
for rep in 1 2; do
  for b in 32 16 64; do
    python3 bench.py --steps 100 --warmup 10 --beam $b --no-cpu-baseline --no-paths 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
p = d['config']['poses']
print('beam tile $b: %.4f ms/step;' % d['ms_per_step'], ' '.join('%s: %.4f in flight, %.4f alone;' % (k, v['ms_per_frame'], v['ms_per_frame_alone']) for k, v in p.items()))" | tee -a $OUT || exit 1
  done
done
