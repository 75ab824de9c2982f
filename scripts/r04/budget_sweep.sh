#!/bin/bash
# The searches' visit budget re-swept on the round's final searches (cheaper per visit): ms per frame, three in flight / alone, poses A B C.
set -o pipefail
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/budget_sweep.txt; : > $OUT
for rep in 1 2; do
  for b in 256 192 384 512; do
    python3 bench.py --steps 100 --warmup 10 --beam-budget $b --no-cpu-baseline --no-paths 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
p = d['config']['poses']
print('budget $b:', ' '.join('%s: %.4f in flight, %.4f alone;' % (k, v['ms_per_frame'], v['ms_per_frame_alone']) for k, v in p.items()))" | tee -a $OUT || exit 1
  done
done
