#!/bin/bash
# The driver's arguments (--steps 20 --warmup 5) with more or fewer untimed settle frames in front: is the short run's lower rate a matter of
# what precedes the timed region (clocks, caches, scheduling state) or of the region's own head and tail?
set -o pipefail
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/settle_probe.txt; : > $OUT
for rep in 1 2; do
  for a in "--steps 20 --warmup 5 --settle 32" "--steps 20 --warmup 5 --settle 128" "--steps 20 --warmup 5 --settle 512" "--steps 20 --warmup 5 --settle 2048" "--steps 200 --warmup 10 --settle 32" "--steps 60 --warmup 5 --settle 32"; do
    python3 bench.py $a --no-cpu-baseline --no-paths --no-poses 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$a', 'Mrays/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'device %.4f' % d['config']['device_ms_per_step'], 'alone %.4f' % d['config']['kernel_ms_alone'])" | tee -a $OUT || exit 1
  done
done
