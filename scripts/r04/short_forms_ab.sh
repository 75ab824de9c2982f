#!/bin/bash
# The driver's arguments under the launch forms 3 (automatic), 0 (always beam kernel then trace kernel) and 2 (always the joint launch), and
# with 2 / 3 / 4 frames in flight: what the first and the last group of a 20-frame region cost under each.
set -o pipefail
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/short_forms_ab.txt; : > $OUT
for rep in 1 2 3; do
  for a in "--fused 3" "--fused 0" "--fused 2" "--fused 2 --frames-in-flight 2" "--fused 3 --frames-in-flight 4"; do
    python3 bench.py --steps 20 --warmup 5 $a --no-cpu-baseline --no-paths --no-poses 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$a', 'Mrays/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'device %.4f' % d['config']['device_ms_per_step'])" | tee -a $OUT || exit 1
  done
done
